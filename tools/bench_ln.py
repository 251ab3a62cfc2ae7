#!/usr/bin/env python3
"""LayerNorm micro-benchmark (M = 7932 rows): rows-per-wave 1 vs 2 via
wn_tune_set("ln_rows")."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wenet_amd import _lib  # noqa: E402

L = _lib.lib()
for M, D in ((7932, 256), (15800, 512)):
    x = torch.randn(M, D, device='cuda')
    w = torch.randn(D, device='cuda')
    b = torch.randn(D, device='cuda')
    y = torch.empty_like(x)
    for rows in (1, 2, 1, 2):
        _lib.check(L.wn_tune_set(b'ln_rows', rows), 'tune')
        for _ in range(5):
            L.wn_op_layernorm(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, D,
                              1e-5, None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            L.wn_op_layernorm(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, D,
                              1e-5, None)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 200
        print(f'M={M} D={D} rows/wave={rows}: {us:.2f} us  {2 * M * D * 4 / us / 1e6:.2f} TB/s')
