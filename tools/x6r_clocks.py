"""Phase stamps of the row-block six-product GEMM (csrc/gemm_x6r.hip, K = 256): shader-clock cycles
of block 0 / wave 0 for the prologue (rows -> LDS, first W loads, barrier), the split into planes,
the main loop and the epilogue, with one block and with the 248 of config 2.
GPU only:  python tools/x6r_clocks.py
"""
import os, sys, ctypes
import torch
sys.path.insert(0, os.getcwd())
from wenet_amd import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
_lib.check(L.wn_tune_set(b'x6_probe', 8), 'tune')
for n, epi in ((768, 0), (256, 1), (256, 0)):
    for M in (32, 7932):
        A = torch.randn(M, 256, device='cuda'); W = torch.randn(n, 256, device='cuda') / 16
        b = torch.randn(n, device='cuda'); x = torch.randn(M, n, device='cuda')
        lw, lb = torch.ones(n, device='cuda'), torch.zeros(n, device='cuda')
        y = torch.empty(M, n, device='cuda'); C = torch.empty(M, n, device='cuda')
        for _ in range(3):
            _lib.check(L.wn_op_gemm_x6r(A.data_ptr(), W.data_ptr(), b.data_ptr(), x.data_ptr(),
                                        lw.data_ptr(), lb.data_ptr(), y.data_ptr(), C.data_ptr(),
                                        M, n, epi, 0.0 if epi else 1.0, 1e-5, 1, st), 'x6r')
        torch.cuda.synchronize()
        out = (ctypes.c_uint64 * 64)()
        _lib.check(L.wn_profile_gemm_clocks(out), 'clk')
        k = list(out)[:5]
        print(f'x6r N={n} epi {epi} M={M}: prologue (rows -> LDS, first W loads, barrier) {k[1]-k[0]}, '
              f'LDS -> split3 {k[2]-k[1]}, main loop {k[3]-k[2]}, epilogue {k[4]-k[3]}, total {k[4]-k[0]} cycles', flush=True)
