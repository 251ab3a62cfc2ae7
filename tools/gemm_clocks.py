"""Shader-clock stamps of the six-product tile GEMM (csrc/gemm_x6.hip, wn_tune_set("x6_probe", 4)):
prologue / K loop / epilogue cycles of one block, cycles per k block (ideal: MFMAs of the SIMD's
waves x 32) and the clock the kernel ran at.  GPU only:  python tools/gemm_clocks.py
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wenet_amd import _lib  # noqa: E402

SHAPES = {'conv2-like (2 rounds of 256-row tiles)': (131072, 256, 2304, 256),
          'big 8192 x 4096 x 4096': (8192, 4096, 4096, 256),
          'sub_out-like K slices': (7936, 256, 4864, 0)}


def main():
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    out = np.zeros((8, 8), dtype=np.uint64)
    for name, (m, n, k, bm) in SHAPES.items():
        A = torch.randn(m, k, device='cuda')
        W = torch.randn(n, k, device='cuda') / k ** 0.5
        b = torch.randn(n, device='cuda')
        C = torch.empty(m, n, device='cuda')
        _lib.check(L.wn_tune_set(b'x6_probe', 4), 'tune')
        for _ in range(3):
            _lib.check(L.wn_op_gemm_x6(A.data_ptr(), W.data_ptr(), b.data_ptr(), None,
                                       C.data_ptr(), m, n, k, 1.0, 0, bm, 1, st), 'x6')
            torch.cuda.synchronize()
        _lib.check(L.wn_profile_gemm_clocks(out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))),
                   'clocks')
        _lib.check(L.wn_tune_set(b'x6_probe', 0), 'tune')
        o = out.astype(np.int64)
        nkb = int(o[0, 6])
        print(f'{name}: M={m} N={n} K={k}, {nkb} k blocks per block')
        for w in range(8):
            if o[w, 3] == 0:
                continue
            pro, loop, epi = o[w, 1] - o[w, 0], o[w, 2] - o[w, 1], o[w, 3] - o[w, 2]
            ns = (o[w, 5] - o[w, 4]) * 10.0
            print(f'  wave {w}: prologue {pro:7d} | loop {loop:8d} = {loop / max(nkb, 1):7.1f} per k block '
                  f'| epilogue {epi:7d} | {ns / 1e3:7.1f} us at {(o[w, 3] - o[w, 0]) / ns:.2f} GHz')
        sys.stdout.flush()


if __name__ == '__main__':
    main()
