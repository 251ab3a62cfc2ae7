#!/usr/bin/env python3
"""tests/golden/chunk_*.npz: the REAL reference's `forward_chunk` API
(BaseEncoder.forward_chunk, encoder.py:204-285 == ASRModel.forward_encoder_chunk,
asr_model.py:385-427) driven chunk by chunk like forward_chunk_by_chunk
(encoder.py:287-362) on seeded synthetic models: every chunk's output and the
attention / convolution caches after chunk `probe` and after the last chunk.
Runs only where /root/reference exists."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gen_golden  # noqa: E402

CASES = [
    dict(case='chunk_tiny_c4', config='tiny_causal', wseed=0, frames=173, fseed=71,
         chunk=4, left=-1, probe=2),
    dict(case='chunk_tiny_c3_l2', config='tiny_causal', wseed=2, frames=131, fseed=72,
         chunk=3, left=2, probe=4),
    dict(case='chunk_tiny_c5_l0', config='tiny_causal', wseed=1, frames=90, fseed=73,
         chunk=5, left=0, probe=1),
    dict(case='chunk_tiny_sym_c6_l1', config='tiny_sym', wseed=0, frames=150, fseed=74,
         chunk=6, left=1, probe=2),
    dict(case='chunk_aishell_c16_l1', config='aishell_u2pp', wseed=0, frames=210,
         fseed=75, chunk=16, left=1, probe=1),
]


def chunk_windows(n_frames, chunk):
    """(start, end) feature windows of forward_chunk_by_chunk for
    Conv2dSubsampling4 (rate 4, right context 6)."""
    context, stride = 7, 4 * chunk
    window = (chunk - 1) * 4 + context
    return [(cur, min(cur + window, n_frames))
            for cur in range(0, n_frames - context + 1, stride)]


def main():
    from wenet_amd import synthetic as S
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    outdir = os.path.join(ROOT, 'tests', 'golden')
    for c in CASES:
        configs = S.make_configs(c['config'])
        sd = S.make_state_dict(configs, c['wseed'])
        model = gen_golden.build_reference_model(configs, sd)
        feats, _ = S.make_features(1, (c['frames'], c['frames']), seed=c['fseed'])
        att = torch.zeros(0, 0, 0, 0)
        cnn = torch.zeros(0, 0, 0, 0)
        outs, offset, arrays = [], 0, {}
        required = c['chunk'] * c['left']
        with torch.no_grad():
            for i, (a, b) in enumerate(chunk_windows(c['frames'], c['chunk'])):
                y, att, cnn = model.forward_encoder_chunk(feats[:, a:b], offset, required,
                                                          att, cnn)
                outs.append(y)
                offset += y.size(1)
                if i == c['probe']:
                    arrays['att_probe'] = att.numpy().copy()
                    arrays['cnn_probe'] = cnn.numpy().copy()
            enc = model.encoder
            if enc.static_chunk_size > 0 or enc.use_dynamic_chunk:
                ref, _ = enc.forward_chunk_by_chunk(feats, c['chunk'], c['left'])
            else:  # forward_chunk itself has no such precondition
                ref = None
        ys = torch.cat(outs, 1)
        assert ref is None or torch.equal(ys, ref)
        arrays['enc_out'] = ys[0].numpy().astype(np.float32)
        arrays['att_last'] = att.numpy().copy()
        arrays['cnn_last'] = cnn.numpy().copy()
        meta = dict(c)
        meta['chunk_sizes'] = [int(y.size(1)) for y in outs]
        path = os.path.join(outdir, c['case'] + '.npz')
        np.savez_compressed(path, meta=np.frombuffer(json.dumps(meta).encode(),
                                                     dtype=np.uint8), **arrays)
        print(path, os.path.getsize(path) // 1024, 'KiB', tuple(ys.shape),
              'att', tuple(att.shape), 'cnn', tuple(cnn.shape))


if __name__ == '__main__':
    main()
