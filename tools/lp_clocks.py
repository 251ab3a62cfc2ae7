"""Shader-clock stamps of the pipelined bf16 / MXFP8 GEMM (csrc/gemm_bf16p.hip,
wn_tune_set("lp_probe", 4)) at the Whisper-large shapes: prologue / K loop / epilogue issue /
store drain cycles of the first, the middle and the last block, and the clock the kernel ran at.
GPU only:  python tools/lp_clocks.py [--lowp bf16|fp8]
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wenet_amd import _lib  # noqa: E402

SHAPES = {  # name: (M, N, K, act, resid, c_lowp)
    'wh_qkv (bf16 C)': (24000, 3840, 1280, 0, False, True),
    'wh_w1 (GELU, C in the operand type)': (24000, 5120, 1280, 3, False, True),
    'wh_w2 (+ residual, fp32 C)': (24000, 1280, 5120, 0, True, False),
    'wh_out (+ residual, fp32 C)': (24000, 1280, 1280, 0, True, False),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lowp', default='bf16', choices=['bf16', 'fp8'])
    args = ap.parse_args()
    L = _lib.lib()
    dev = torch.device('cuda', 0)
    out = np.zeros((8, 8), dtype=np.uint64)
    for name, (m, n, k, act, resid, c_lowp) in SHAPES.items():
        if args.lowp == 'fp8' and 'w' not in name.split()[0][3:]:
            continue                      # fp8 mode: only the FFN GEMMs are MXFP8
        g = torch.Generator().manual_seed(1)
        A = (torch.rand(m, k, generator=g) * 2 - 1).to(dev)
        W = ((torch.rand(n, k, generator=g) * 2 - 1) * 0.1).to(dev)
        bias = torch.rand(n, generator=g).to(dev)
        R = torch.rand(m, n, generator=g).to(dev) if resid else None
        if args.lowp == 'bf16':
            A2, W2, sa, sw, dt = A.to(torch.bfloat16), W.to(torch.bfloat16), None, None, 1
            cmode = 1 if c_lowp else 0
            C = torch.empty(m, n, device=dev, dtype=torch.bfloat16 if c_lowp else torch.float32)
        else:
            def mxq(x):
                r, kk = x.shape
                q = torch.empty((r, kk), dtype=torch.uint8, device=dev)
                sc = torch.zeros((kk // 128, r), dtype=torch.int32, device=dev)
                _lib.check(L.wn_op_mx_quantize(x.data_ptr(), r, kk, q.data_ptr(), sc.data_ptr(),
                                               None), 'mxq')
                return q, sc
            (A2, sa), (W2, sw) = mxq(A), mxq(W)
            dt = 2
            cmode = 2 if c_lowp else 0
            C = torch.empty(m, n, device=dev, dtype=torch.uint8 if c_lowp else torch.float32)
        csc = torch.zeros(((n + 127) // 128, m), dtype=torch.int32, device=dev)

        def run():
            _lib.check(L.wn_op_gemm_lowp(A2.data_ptr(), W2.data_ptr(),
                                         sa.data_ptr() if sa is not None else None,
                                         sw.data_ptr() if sw is not None else None,
                                         bias.data_ptr(), R.data_ptr() if resid else None,
                                         C.data_ptr(), csc.data_ptr(), m, n, k, 1.0, act, cmode, dt,
                                         None), name)
        _lib.check(L.wn_tune_set(b'gemm_tile_bf16', 8), 'tune')
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100.0
        tiles = -(-m // 256) * -(-n // 256)
        print(f'{name} [{args.lowp}]: M={m} N={n} K={k}: {us:.1f} us = {2e-6 * m * n * k / us:.0f} TF/s, '
              f'{tiles} tiles = {tiles / 256:.2f} rounds')
        for probe, which in ((12, 'first block'), (4, 'middle block'), (20, 'last block')):
            _lib.check(L.wn_tune_set(b'lp_probe', probe), 'tune')
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            _lib.check(L.wn_profile_gemm_clocks(out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))),
                       'clocks')
            _lib.check(L.wn_tune_set(b'lp_probe', 0), 'tune')
            o = out.astype(np.int64)
            nk = int(o[0, 7])
            for w in (0, 4):
                pro, loop, epi, drain = (o[w, 1] - o[w, 0], o[w, 2] - o[w, 1], o[w, 3] - o[w, 2],
                                         o[w, 4] - o[w, 3])
                ns = (o[w, 6] - o[w, 5]) * 10.0
                print(f'  {which:12s} wave {w}: prologue {pro:6d} | loop {loop:7d} = {loop / max(nk, 1):6.1f} per '
                      f'K tile ({nk}) | epilogue issue {epi:6d} | drain {drain:6d} | {ns / 1e3:6.1f} us at '
                      f'{(o[w, 4] - o[w, 0]) / max(ns, 1):.2f} GHz')
        sys.stdout.flush()
    L.wn_tune_set(b'gemm_tile_bf16', 0)


if __name__ == '__main__':
    main()
