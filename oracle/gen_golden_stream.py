#!/usr/bin/env python3
"""tests/golden/stream_*.npz: the REAL reference's cache-based streaming path
(`ASRModel.decode(..., simulate_streaming=True)` ->
BaseEncoder.forward_chunk_by_chunk, encoder.py:287-362) on seeded synthetic
models, one utterance per case.  Runs only where /root/reference exists."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gen_golden  # noqa: E402

CASES = [
    dict(case='stream_tiny_c4', config='tiny_causal', wseed=0, frames=173, fseed=61,
         chunk=4, left=-1, beam=4, ctc_weight=0.5, reverse_weight=0.3),
    dict(case='stream_tiny_c3_l2', config='tiny_causal', wseed=2, frames=131, fseed=62,
         chunk=3, left=2, beam=3, ctc_weight=0.3, reverse_weight=0.0),
    dict(case='stream_aishell_c16', config='aishell_u2pp', wseed=0, frames=333,
         fseed=63, chunk=16, left=-1, beam=10, ctc_weight=0.5, reverse_weight=0.3),
]


def main():
    from wenet_amd import synthetic as S
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    outdir = os.path.join(ROOT, 'tests', 'golden')
    methods = ['ctc_greedy_search', 'ctc_prefix_beam_search', 'attention_rescoring']
    for c in CASES:
        configs = S.make_configs(c['config'])
        sd = S.make_state_dict(configs, c['wseed'])
        model = gen_golden.build_reference_model(configs, sd)
        feats, lens = S.make_features(1, (c['frames'], c['frames']), seed=c['fseed'])
        with torch.no_grad():
            enc, mask = model.encoder.forward_chunk_by_chunk(feats, c['chunk'], c['left'])
            res = model.decode(methods, feats, lens, beam_size=c['beam'],
                               decoding_chunk_size=c['chunk'],
                               num_decoding_left_chunks=c['left'],
                               ctc_weight=c['ctc_weight'],
                               reverse_weight=c['reverse_weight'],
                               simulate_streaming=True)
        meta = dict(c)
        meta['greedy'] = res['ctc_greedy_search'][0].tokens
        p = res['ctc_prefix_beam_search'][0]
        meta['prefix'] = dict(nbest=[list(map(int, h)) for h in p.nbest],
                              nbest_scores=[float(s) for s in p.nbest_scores],
                              nbest_times=[list(map(int, t)) for t in p.nbest_times])
        r = res['attention_rescoring'][0]
        meta['rescoring'] = dict(tokens=list(map(int, r.tokens)), score=float(r.score))
        path = os.path.join(outdir, c['case'] + '.npz')
        np.savez_compressed(path, enc_out=enc[0].numpy().astype(np.float32),
                            meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
        print(path, os.path.getsize(path) // 1024, 'KiB', tuple(enc.shape))


if __name__ == '__main__':
    main()
