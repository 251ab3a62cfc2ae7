"""Stand-alone timing of the row-block six-product GEMMs: csrc/gemm_x6r.hip at the config-2
shapes (K = 256: out-projection / pointwise_conv2 + residual + LayerNorm, QKV) and
csrc/gemm_x6r512.hip at the config-3 / config-4 shapes (K = 512: projections of one to three
512-column passes, + residual + LayerNorm, + the pointwise_conv1 + GLU chain).
GPU only:  python tools/bench_x6r.py [knob=value ...]   (knobs go to wn_tune_set first)
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
    for kv in sys.argv[1:]:
        k, v = kv.split('=')
        _lib.check(L.wn_tune_set(k.encode(), int(v)), 'wn_tune_set ' + kv)
        print('knob', kv, flush=True)
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
    for m in (7932, 16231):
        for n, epi, what in ((512, 0, 'proj'), (1024, 0, 'proj'), (1536, 0, 'QKV'),
                             (512, 1, 'proj + residual + LayerNorm'),
                             (512, 3, 'proj + residual + LayerNorm + pointwise_conv1 + GLU')):
            A = torch.randn(m, 512, device='cuda')
            W = torch.randn(n, 512, device='cuda') / 22
            W2 = torch.randn(1024, 512, device='cuda') / 22
            b, b2 = torch.randn(n, device='cuda'), torch.randn(1024, device='cuda')
            x = torch.randn(m, n, device='cuda')
            lw, lb = torch.ones(n, device='cuda'), torch.zeros(n, device='cuda')
            y = torch.empty(m, n, device='cuda')
            C = torch.empty(m, n, device='cuda')

            def run5(r):
                _lib.check(L.wn_op_gemm_x6r512(A.data_ptr(), W.data_ptr(), b.data_ptr(),
                                               x.data_ptr(), lw.data_ptr(), lb.data_ptr(),
                                               y.data_ptr(), W2.data_ptr(), b2.data_ptr(),
                                               C.data_ptr(), m, n, epi, 0.0 if epi else 1.0, 1e-5,
                                               r, st), 'x6r512')
            run5(1)
            best = 1e9
            for _ in range(3):
                t1 = timed(lambda: run5(1))
                tn = timed(lambda: run5(reps + 1))
                best = min(best, (tn - t1) / reps * 1e3)
            mf = (n + (1024 if epi == 3 else 0)) / 32 / 4 * 32 * 6     # MFMAs per wave
            print(f'x6r512 M={m} N={n} epi {epi} ({what}): {best:7.1f} us  '
                  f'({mf:.0f} MFMAs per wave = {mf * 32 / 2.1e3:.1f} us at 2.1 GHz; W image '
                  f'{(n + (1024 if epi == 3 else 0)) * 512 * 6 / 1e6:.1f} MB per block)', flush=True)


if __name__ == '__main__':
    main()
