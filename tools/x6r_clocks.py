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
        k = list(out)[8 * epi:8 * epi + 8]
        print(f'x6r N={n} epi {epi} M={M}: rows in registers {k[5]-k[0]}, -> LDS + first W loads + barrier '
              f'{k[1]-k[5]}, split3 -> planes {k[2]-k[1]}, main loop {k[3]-k[2]}, epilogue {k[4]-k[3]}, '
              f'total {k[4]-k[0]} cycles in {(k[7]-k[6]) * 10} ns', flush=True)
if len(sys.argv) > 1 and sys.argv[1] == 'encoder':
    # the kernels as the encoder launches them (config 2's batch): row [epi + 4 (prologue form)]
    from wenet_amd import synthetic as S
    from wenet_amd.model import ASRModel
    configs = S.make_configs('aishell_u2pp')
    model = ASRModel(configs, S.make_state_dict(configs, 0), device='cuda:0')
    feats, lens = S.make_bench_batch('config2', 1)
    for _ in range(2):
        model._forward_encoder(feats.cuda(), lens, -1, -1)
    torch.cuda.synchronize()
    out = (ctypes.c_uint64 * 64)()
    _lib.check(L.wn_profile_gemm_clocks(out), 'clk')
    names = {4: 'QKV with the FFN-partials prologue (epi 0, PRO 1)', 3: 'out-proj + LN + pw1 + GLU chain (epi 3)',
             5: 'dwconv prologue + pointwise_conv2 + LN (epi 1, PRO 2)'}
    for row, what in names.items():
        k = list(out)[8 * row:8 * row + 8]
        print(f'{what}: rows in registers {k[5]-k[0]}, -> LDS + first W loads + barrier {k[1]-k[5]}, '
              f'split3 -> planes {k[2]-k[1]}, main loop {k[3]-k[2]}, epilogue {k[4]-k[3]}, total {k[4]-k[0]} cycles in {(k[7]-k[6]) * 10} ns')
