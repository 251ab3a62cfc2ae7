#!/usr/bin/env python3
"""Debug probe: decode() n-best scores of a bench batch with conv2 on the six-product GEMM vs
v_mfma_f32, repeated, in both orders."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from wenet_amd import _lib, synthetic as S  # noqa: E402
from wenet_amd.model import ASRModel  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else 'config3'
methods = sys.argv[2].split(',') if len(sys.argv) > 2 else ['ctc_prefix_beam_search']
w = S.BENCH_WORKLOADS[wl]
configs = S.make_configs(w['config'])
model = ASRModel(configs, S.make_state_dict(configs, 0), device='cuda:0')
feats, lens = S.make_bench_batch(wl, 1)
L = _lib.lib()
fd = feats.cuda()
ref = None
for conv in (1, 1, 0, 0, 1):
    _lib.check(L.wn_tune_set(b'x6_conv', conv), 'tune')
    res = model.decode(methods, fd, lens, beam_size=10, **w['kw'])['ctc_prefix_beam_search']
    sc = [r.nbest_scores[0] for r in res]
    if conv == 0 and ref is None:
        ref = sc
    print('conv', conv, 'utt 17..21', [round(x, 4) for x in sc[17:22]])
    if ref is not None:
        bad = [(i, round(a - b, 4)) for i, (a, b) in enumerate(zip(sc, ref)) if abs(a - b) > 1e-3]
        print('   vs f32 conv: utterances off by > 1e-3:', bad)
