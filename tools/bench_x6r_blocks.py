"""Row-block six-product GEMM (csrc/gemm_x6r.hip) as a function of the NUMBER OF BLOCKS (M = 32 ..
7932 rows): is a launch bound by what one block does or by what all of them share?
GPU only:  python tools/bench_x6r_blocks.py
"""
import os, sys, time
import torch
sys.path.insert(0, os.getcwd())
from wenet_amd import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
reps = 200
for n, epi in ((768, 0), (256, 1), (256, 0)):
    for M in (32, 64, 256, 992, 1984, 3968, 7932):
        A = torch.randn(M, 256, device='cuda'); W = torch.randn(n, 256, device='cuda') / 16
        b = torch.randn(n, device='cuda'); x = torch.randn(M, n, device='cuda')
        lw, lb = torch.ones(n, device='cuda'), torch.zeros(n, device='cuda')
        y = torch.empty(M, n, device='cuda'); C = torch.empty(M, n, device='cuda')
        def run(r):
            _lib.check(L.wn_op_gemm_x6r(A.data_ptr(), W.data_ptr(), b.data_ptr(), x.data_ptr(),
                                        lw.data_ptr(), lb.data_ptr(), y.data_ptr(), C.data_ptr(),
                                        M, n, epi, 0.0 if epi else 1.0, 1e-5, r, st), 'x6r')
        run(1); best = 1e9
        for _ in range(3):
            t1 = timed(lambda: run(1)); tn = timed(lambda: run(reps + 1))
            best = min(best, (tn - t1) / reps * 1e3)
        print(f'x6r N={n} epi {epi} M={M:5d} ({(M + 31) // 32:3d} blocks): {best:6.1f} us', flush=True)
