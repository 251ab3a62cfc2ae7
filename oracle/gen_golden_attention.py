#!/usr/bin/env python3
"""tests/golden/attn_*.npz: the REAL reference's `attention` decode mode
(attention_beam_search, search.py:252-371, with the decoder caches of
decoder.py:226-281) on seeded synthetic models.  Runs only where /root/reference
exists."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gen_golden  # noqa: E402

CASES = [
    dict(case='attn_tiny_causal', config='tiny_causal', wseed=0, batch=3,
         frames=(60, 150), fseed=10, beam=4, length_penalty=0.0),
    dict(case='attn_tiny_sym_lp', config='tiny_sym', wseed=1, batch=4,
         frames=(40, 120), fseed=11, beam=3, length_penalty=0.5),
    dict(case='attn_aishell', config='aishell_u2pp', wseed=0, batch=2,
         frames=(120, 160), fseed=12, beam=10, length_penalty=0.0),
]


def main():
    from wenet_amd import synthetic as S
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    outdir = os.path.join(ROOT, 'tests', 'golden')
    for c in CASES:
        configs = S.make_configs(c['config'])
        sd = S.make_state_dict(configs, c['wseed'])
        model = gen_golden.build_reference_model(configs, sd)
        feats, lens = S.make_features(c['batch'], c['frames'], seed=c['fseed'])
        with torch.no_grad():
            res = model.decode(['attention'], feats, lens, beam_size=c['beam'],
                               length_penalty=c['length_penalty'])['attention']
        meta = dict(c)
        meta['tokens'] = [list(map(int, r.tokens)) for r in res]
        path = os.path.join(outdir, c['case'] + '.npz')
        np.savez_compressed(path, meta=np.frombuffer(json.dumps(meta).encode(),
                                                     dtype=np.uint8))
        print(path, [len(t) for t in meta['tokens']])


if __name__ == '__main__':
    main()
