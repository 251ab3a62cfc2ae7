"""Stand-alone timing of the row-block six-product GEMM (csrc/gemm_x6r.hip) at the config-2
shapes: out-projection / pointwise_conv2 + residual + LayerNorm (N = 256) and QKV (N = 768).
GPU only:  python tools/bench_x6r.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wenet_amd import _lib  # noqa: E402

M = 7932


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


def main():
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    reps = 50
    for n, epi, what in ((256, 1, 'proj + residual + LayerNorm'), (256, 0, 'proj'),
                         (512, 0, 'proj'), (512, 2, 'pointwise_conv1 + GLU'), (768, 0, 'QKV')):
        A = torch.randn(M, 256, device='cuda')
        W = torch.randn(n, 256, device='cuda') / 16
        b = torch.randn(n, device='cuda')
        x = torch.randn(M, n, device='cuda')
        lw, lb = torch.ones(n, device='cuda'), torch.zeros(n, device='cuda')
        y = torch.empty(M, n, device='cuda')
        C = torch.empty(M, n, device='cuda')

        def run(r):
            _lib.check(L.wn_op_gemm_x6r(A.data_ptr(), W.data_ptr(), b.data_ptr(), x.data_ptr(),
                                        lw.data_ptr(), lb.data_ptr(), y.data_ptr(), C.data_ptr(),
                                        M, n, epi, 0.0 if epi else 1.0, 1e-5, r, st), 'x6r')
        run(1)
        best = 1e9
        for _ in range(3):
            t1 = timed(lambda: run(1))
            tn = timed(lambda: run(reps + 1))
            best = min(best, (tn - t1) / reps * 1e3)
        print(f'x6r M={M} N={n} K=256 epi {epi} ({what}): {best:7.1f} us', flush=True)


if __name__ == '__main__':
    main()
