#!/usr/bin/env python3
"""GEMM micro-benchmark over the shapes of BASELINE config 2 (AIShell 256d,
B=32 x ~10 s -> M = 7932 encoder rows), through the C ABI (wn_op_gemm).

    python tools/bench_gemm.py [--reps 30] [--only w1] [--variants 0,1]
    python tools/bench_gemm.py --bf16 --tiles 0,1,7 --only wh_w1,wh_w2,w1

Prints TFLOP/s per shape and tile rule (gemm_tile_bf16; the run-time epilogue variants were removed in round 4),
interleaved within one process.
"""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wenet_amd import _lib  # noqa: E402

M = 7932
SHAPES = {
    # name: (M, N, K, act, resid)
    'w1': (M, 2048, 256, 1, False),
    'w2': (M, 256, 2048, 0, True),
    'qkv': (M, 768, 256, 0, False),
    'out': (M, 256, 256, 0, True),
    'ctc': (M, 4233, 256, 0, False),
    'sub_out': (M, 256, 4864, 0, False),
    'w1_512': (M, 2048, 512, 1, False),
    'w2_512': (M, 512, 2048, 0, True),
    # Whisper-large-v3 encoder, batch 16 x 1500 frames (BASELINE configs[4])
    'wh_w1': (24000, 5120, 1280, 3, False),
    'wh_w2': (24000, 1280, 5120, 0, True),
    'wh_qkv': (24000, 3840, 1280, 0, False),
    'wh_out': (24000, 1280, 1280, 0, True),
    'wh_w2n': (24000, 1280, 5120, 0, False),   # w_2 without the residual (variants)
    'sq8k': (8192, 8192, 8192, 0, False),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--only', default='')
    ap.add_argument('--variants', default='0')
    ap.add_argument('--tiles', default='0')
    ap.add_argument('--bf16', action='store_true',
                    help='bf16-operand kernels (wn_op_gemm_bf16, gemm_tile_bf16)')
    ap.add_argument('--lowp', default='', choices=['', 'bf16', 'fp8'],
                    help='operands already stored as bf16 / e4m3 (wn_op_gemm_lowp): the '
                         'kernels the bf16 / fp8 modes run; tiles via gemm_tile_bf16 '
                         '(1 = 128x128, 7 = 256x256 register-staged, 8 = 256x256 pipelined)')
    ap.add_argument('--c-bf16', action='store_true',
                    help='--lowp: C in the operand type (bf16 / MXFP8; no residual)')
    args = ap.parse_args()
    L = _lib.lib()
    op = L.wn_op_gemm_bf16 if args.bf16 else L.wn_op_gemm
    tile_key = b'gemm_tile_bf16' if (args.bf16 or args.lowp) else b'gemm_tile'
    dev = torch.device('cuda', 0)
    variants = [(int(t), int(v)) for t in args.tiles.split(',')
                for v in args.variants.split(',')]
    names = [n for n in SHAPES if not args.only or n in args.only.split(',')]
    out = {}
    for name in names:
        m, n, k, act, resid = SHAPES[name]
        g = torch.Generator().manual_seed(1)
        A = (torch.rand(m, k, generator=g) * 2 - 1).to(dev)
        W = ((torch.rand(n, k, generator=g) * 2 - 1) * 0.1).to(dev)
        bias = torch.rand(n, generator=g).to(dev)
        R = torch.rand(m, n, generator=g).to(dev) if resid else None
        C = torch.empty(m, n, device=dev)
        if args.lowp:
            if args.c_bf16:
                resid, R = False, None
            C = torch.empty(m, n, device=dev,
                            dtype=torch.bfloat16 if args.c_bf16 else torch.float32)
            if args.lowp == 'bf16':
                A2, W2, sa, sw, dt = A.to(torch.bfloat16), W.to(torch.bfloat16), None, None, 1
                cmode = 1 if args.c_bf16 else 0
            else:
                def mxq(x):
                    r, kk = x.shape
                    q = torch.empty((r, kk), dtype=torch.uint8, device=dev)
                    sc = torch.zeros((kk // 128, r), dtype=torch.int32, device=dev)
                    _lib.check(L.wn_op_mx_quantize(x.data_ptr(), r, kk, q.data_ptr(),
                                                   sc.data_ptr(), None), 'mxq')
                    return q, sc
                (A2, sa), (W2, sw) = mxq(A), mxq(W)
                dt = 2
                cmode = 2 if args.c_bf16 else 0
                if args.c_bf16:
                    C = torch.empty(m, n, device=dev, dtype=torch.uint8)
            csc = torch.zeros(((n + 127) // 128, m), dtype=torch.int32, device=dev)

        def run_lowp():
            _lib.check(L.wn_op_gemm_lowp(A2.data_ptr(), W2.data_ptr(),
                                         sa.data_ptr() if sa is not None else None,
                                         sw.data_ptr() if sw is not None else None,
                                         bias.data_ptr(), R.data_ptr() if resid else None,
                                         C.data_ptr(), csc.data_ptr(), m, n, k, 1.0, act,
                                         cmode, dt, None), name)

        def run():
            if args.lowp:
                return run_lowp()
            _lib.check(op(A.data_ptr(), W.data_ptr(), bias.data_ptr(),
                          R.data_ptr() if resid else None,
                          C.data_ptr(), m, n, k, 1.0, act, None), name)
        res = {}
        for rnd in range(3):
            for v in variants:
                L.wn_tune_set(tile_key, v[0])
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    run()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / args.reps
                res.setdefault(v, []).append(us)
        for v in variants:
            us = min(res[v])
            tf = 2.0 * m * n * k / us / 1e6
            print(f'{name:8s} M={m} N={n} K={k} tile/variant {v}: {us:8.2f} us  '
                  f'{tf:6.1f} TF/s  (rounds {[round(x, 1) for x in res[v]]})', flush=True)
            out[f'{name}/t{v[0]}v{v[1]}'] = dict(us=us, tflops=tf)
    L.wn_tune_set(tile_key, 0)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
