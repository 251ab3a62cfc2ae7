"""Shader-clock stamps of the fused six-product feed-forward kernel (csrc/ffn_x6f.hip,
VAR & 8192): cycles per sub-stage (48 MFMAs = 1536 matrix-pipe cycles) of one block's last
steady-state chunk, for the default kernel and its ablations.  GPU only:
    python tools/ffn_clocks.py
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wenet_amd import _lib  # noqa: E402

M, D, F = 7932, 256, 2048
KINDS = ('A', 'A+sync', 'A piece', 'A piece+sync', 'B piece', 'B piece+sync', 'B', 'B+sync')


def main():
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    X = torch.randn(M, D, device='cuda')
    W1 = torch.randn(F, D, device='cuda') / D ** 0.5
    W2 = torch.randn(D, F, device='cuda') / F ** 0.5
    b1, b2 = torch.randn(F, device='cuda'), torch.randn(D, device='cuda')
    x = torch.randn(M, D, device='cuda')
    lw, lb = torch.ones(D, device='cuda'), torch.zeros(D, device='cuda')
    y = torch.empty(M, D, device='cuda')
    _lib.check(L.wn_tune_set(b'ffn_x6f', 1), 'tune')
    # FFN_XIMG=1: X handed to the kernel as its plane image (round 5) instead of fp32 rows
    if os.environ.get('FFN_XIMG') == '1':
        _lib.check(L.wn_tune_set(b'ffn_ximg', 2), 'tune')
    out = np.zeros((4, 24), dtype=np.uint64)
    for var, what in ((8704, 'stage DMA as one burst behind the barrier (r03 first form)'), (25088, 'default kernel (DMA spread over the stage)'), (90624, 'three of the six products'), (8768, 'no fragment reads'),
                      (8708, 'no pieces'), (8706, 'no DMA'), (8782, 'MFMAs only')):
        if L.wn_tune_set(b'ffn_x6f_var', var) != 0:
            continue          # every variant but 25088: WN_ABLATION builds only
        rows = []
        for _ in range(5):
            _lib.check(L.wn_op_ffn_x6(X.data_ptr(), W1.data_ptr(), b1.data_ptr(), W2.data_ptr(),
                                      b2.data_ptr(), x.data_ptr(), lw.data_ptr(), lb.data_ptr(),
                                      y.data_ptr(), M, D, F, 1, 0.5, 1e-5, 1, st), 'ffn_x6')
            torch.cuda.synchronize()
            _lib.check(L.wn_profile_ffn_clocks(out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))),
                       'clocks')
            rows.append(np.diff(out[:, :9].astype(np.int64), axis=1))
        d = np.median(np.stack(rows), axis=0)          # [wave][sub]
        print(f'var {var} ({what}): cycles per sub-stage (ideal 1536), median of 5 launches')
        print('  kinds : ' + ' | '.join(KINDS))
        for w in range(4):
            print(f'  wave {w}: ' + ' '.join(f'{int(v):6d}' for v in d[w]) +
                  f' | chunk {int(d[w].sum())}')
        o = out.astype(np.int64)
        cyc = o[:, 20] - o[:, 9]
        ns = (o[:, 13] - o[:, 12]) * 10.0
        print('  prologue of wave 0: entry -> loads + DMA issued %d -> split done %d -> all landed %d -> '
              'bias stored + barrier %d -> loop %d cycles' % tuple(
                  int(o[0, b] - o[0, a]) for a, b in ((9, 15), (15, 16), (16, 17), (17, 18), (18, 10))))
        print('  last launch, per wave: prologue %s | loop %s | epilogue %s cycles; kernel %s cycles in '
              '%s ns -> %s GHz' % ((o[:, 10] - o[:, 9]).tolist(), (o[:, 11] - o[:, 10]).tolist(),
                                   (o[:, 20] - o[:, 11]).tolist(), cyc.tolist(), ns.tolist(),
                                   np.round(cyc / ns, 2).tolist()))
        sys.stdout.flush()
    _lib.check(L.wn_tune_set(b'ffn_x6f_var', 0), 'tune')


if __name__ == '__main__':
    main()
