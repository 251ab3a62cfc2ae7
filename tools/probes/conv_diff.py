#!/usr/bin/env python3
"""Debug probe: subsampling output (encoder after 0 layers) of a bench batch with conv2 on
the six-product GEMM vs on v_mfma_f32 -- which packed rows differ, and where they sit."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from wenet_amd import _lib, synthetic as S  # noqa: E402
from wenet_amd.model import ASRModel  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else 'config3'
w = S.BENCH_WORKLOADS[wl]
configs = S.make_configs(w['config'])
model = ASRModel(configs, S.make_state_dict(configs, 0), device='cuda:0')
feats, lens = S.make_bench_batch(wl, 1)
L = _lib.lib()
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 0
_lib.check(L.wn_debug_set(model._h, b'n_layers', nl), 'dbg')
_lib.check(L.wn_debug_set(model._h, b'skip_after_norm', 1 if nl >= 0 else 0), 'dbg')
outs = {}
for conv in (1, 0):
    _lib.check(L.wn_tune_set(b'x6_conv', conv), 'tune')
    enc, mask = model._forward_encoder(feats.cuda(), lens, -1, -1)
    outs[conv] = enc.cpu()
d = (outs[1] - outs[0]).abs()           # (B, T', d)
print('max diff', float(d.max()))
bad = (d.amax(-1) > 1e-3)
print('bad frames per utterance', bad.sum(1).tolist())
for b in range(bad.shape[0]):
    idx = torch.nonzero(bad[b]).flatten().tolist()
    if idx:
        print('utt', b, 'len', int(lens[b]), 'T2', int(mask[b].sum()) if mask is not None else None,
              'bad t2', idx[:20], '...', idx[-5:], 'maxdiff', float(d[b].max()))
