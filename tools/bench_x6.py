#!/usr/bin/env python3
"""Micro-benchmark of the six-product fp32 GEMM (csrc/gemm_x6.hip) through the C ABI, next
to the v_mfma_f32 kernel (wn_op_gemm) on the shapes of BASELINE config 2.

    python tools/bench_x6.py [--reps 20] [--only w1,w2]

GEMM rows time `reps` launches inside ONE wn_op_gemm_x6 call (the operand split runs
once, outside the repeated part, and is subtracted via a reps = 1 call); FFN rows time the
whole module (split of X, both GEMMs; the reduce + LayerNorm once).
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wenet_amd import _lib  # noqa: E402

M = 7932
SHAPES = {
    'w1': (M, 2048, 256, 1), 'w2': (M, 256, 2048, 0), 'qkv': (M, 768, 256, 0),
    'out': (M, 256, 256, 0), 'ctc': (M, 4236, 256, 0), 'sub_out': (M, 256, 4864, 0),
    'big': (8192, 4096, 4096, 0),
}


def timed(fn, n=3):
    best = 1e9
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--only', default='')
    ap.add_argument('--af32', type=int, default=0,
                    help='1: A as plain fp32 rows split in registers, 0: plane image')
    ap.add_argument('--probe', type=int, default=0,
                    help='ablation: 1 = no MFMAs (DMA + fragment reads), 2 = no DMA')
    args = ap.parse_args()
    L = _lib.lib()
    _lib.check(L.wn_tune_set(b'x6_probe', args.probe), 'tune')
    _lib.check(L.wn_tune_set(b'x6_af32', args.af32), 'tune')
    st = torch.cuda.current_stream().cuda_stream
    only = [s for s in args.only.split(',') if s]
    for name, (m, n, k, act) in SHAPES.items():
        if only and name not in only:
            continue
        A = torch.randn(m, k, device='cuda')
        W = torch.randn(n, k, device='cuda') / k ** 0.5
        b = torch.randn(n, device='cuda')
        C = torch.empty(m, n, device='cuda')
        flops = 2.0 * m * n * k

        def x6(bm, reps):
            _lib.check(L.wn_op_gemm_x6(A.data_ptr(), W.data_ptr(), b.data_ptr(), None,
                                       C.data_ptr(), m, n, k, 1.0, act, bm, reps, st), 'x6')

        def f32():
            for _ in range(args.reps):
                _lib.check(L.wn_op_gemm(A.data_ptr(), W.data_ptr(), b.data_ptr(), None,
                                        C.data_ptr(), m, n, k, 1.0, act, st), 'gemm')
        x6(0, 1); f32()
        row = [f'{name:8s} M={m} N={n} K={k}']
        for bm in (256, 128, 120):
            t1 = timed(lambda: x6(bm, 1))
            tn = timed(lambda: x6(bm, args.reps + 1))
            us = (tn - t1) / args.reps * 1e3
            name = {256: '256x256 8w', 128: '128x256 4w', 120: '128x256 8w'}.get(bm, str(bm))
            row.append(f'x6 {name}: {us:8.1f} us {flops / us / 1e6:7.1f} TF-eq')
        us = timed(f32) / args.reps * 1e3
        row.append(f'f32 mfma: {us:8.1f} us {flops / us / 1e6:7.1f} TF')
        print(' | '.join(row), flush=True)
    if not only or 'ffn' in only:
        for (m, d, f) in ((M, 256, 2048), (16231, 512, 2048)):
            X = torch.randn(m, d, device='cuda')
            W1 = torch.randn(f, d, device='cuda') / d ** 0.5
            W2 = torch.randn(d, f, device='cuda') / f ** 0.5
            b1, b2 = torch.randn(f, device='cuda'), torch.randn(d, device='cuda')
            x = torch.randn(m, d, device='cuda')
            lw, lb = torch.ones(d, device='cuda'), torch.zeros(d, device='cuda')
            y = torch.empty(m, d, device='cuda')

            def ffn(reps):
                _lib.check(L.wn_op_ffn_x6(X.data_ptr(), W1.data_ptr(), b1.data_ptr(),
                                          W2.data_ptr(), b2.data_ptr(), x.data_ptr(),
                                          lw.data_ptr(), lb.data_ptr(), y.data_ptr(), m, d, f,
                                          1, 0.5, 1e-5, reps, st), 'ffn_x6')

            def fused():
                for _ in range(args.reps):
                    _lib.check(L.wn_op_ffn_fused(X.data_ptr(), W1.data_ptr(), b1.data_ptr(),
                                                 W2.data_ptr(), b2.data_ptr(), x.data_ptr(),
                                                 lw.data_ptr(), lb.data_ptr(), y.data_ptr(), m,
                                                 d, f, 1, 0.5, 1e-5, st), 'ffn_fused')
            _lib.check(L.wn_tune_set(b'ffn_x6f', 0), 'tune')
            ffn(1); fused()
            t1 = timed(lambda: ffn(1))
            tn = timed(lambda: ffn(args.reps + 1))
            us = (tn - t1) / args.reps * 1e3
            if d == 256:
                # hidden tensor on chip (csrc/ffn_x6f.hip); includes the split of X
                for ring in (3, 4, 5, 6):
                    _lib.check(L.wn_tune_set(b'ffn_x6f', 1), 'tune')
                    if L.wn_tune_set(b'ffn_x6f_ring', ring) != 0 and ring != 3:
                        continue      # rings 4..6: WN_ABLATION builds only
                    ffn(1)
                    t1f = timed(lambda: ffn(1))
                    tnf = timed(lambda: ffn(args.reps + 1))
                    usf = (tnf - t1f) / args.reps * 1e3
                    print(f'ffn M={m} D={d} F={f} | x6 ON CHIP ring {ring} (fp32 X split in registers): '
                          f'{usf:8.1f} us {4.0 * m * d * f / usf / 1e6:7.1f} TF-eq', flush=True)
                L.wn_tune_set(b'ffn_x6f_ring', 3)
                # measurement variants of the kernel (results wrong by design except 16)
                for var, what in ((82432, 'three of the six products (wrong results: what half the MFMAs would cost)'), (512, 'full, stage DMA as one burst behind the barrier'), (1, 'no MFMAs'),
                                  (2, 'no DMA in the loop'), (4, 'no bias/act/split pieces'), (64, 'no fragment reads'), (10, 'no DMA, no waits'), (78, 'MFMAs only'), (74, 'MFMAs + pieces'), (14, 'MFMAs + fragment reads'), (76, 'MFMAs + DMA'), (70, 'MFMAs + waits/barriers'), (1094, 'MFMAs + s_barrier only'), (4096, 'full, fragment reads of a group as one burst'), (4110, 'MFMAs + burst fragment reads'), (2118, 'MFMAs + waitcnts only'), (128, 'partials stored sc0 sc1'),
                                  (8, 'no waits / barriers in the loop')):
                    if L.wn_tune_set(b'ffn_x6f_var', var) != 0:
                        continue      # WN_ABLATION builds only
                    ffn(1)
                    t1f = timed(lambda: ffn(1))
                    tnf = timed(lambda: ffn(args.reps + 1))
                    usf = (tnf - t1f) / args.reps * 1e3
                    print(f'ffn M={m} D={d} F={f} | x6 ON CHIP var {var:2d} ({what}): '
                          f'{usf:8.1f} us', flush=True)
                _lib.check(L.wn_tune_set(b'ffn_x6f_var', 0), 'tune')
            uf = timed(fused) / args.reps * 1e3
            fl = 4.0 * m * d * f
            print(f'ffn M={m} D={d} F={f} | x6 (split + 2 GEMMs): {us:8.1f} us '
                  f'{fl / us / 1e6:7.1f} TF-eq | fused f32 (+ reduce/LN): {uf:8.1f} us '
                  f'{fl / uf / 1e6:7.1f} TF', flush=True)


if __name__ == '__main__':
    main()
