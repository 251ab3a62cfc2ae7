#!/usr/bin/env python3
"""Generate tests/golden/whisperenc_*.npz: the REAL reference's
TransformerEncoder (wenet/models/transformer/encoder.py:365-437, configured like
examples/aishell/whisper/conf/finetune_whisper_largev3.yaml: conv1d2 subsampling,
abs_pos_whisper, gelu, key_bias=False) + CTC head + search.py on seeded synthetic
weights and log-mel-like inputs.  Runs only where /root/reference exists.

    python oracle/gen_golden_whisper.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import _ref_harness  # noqa: E402

CASES = [
    dict(case='whisperenc_tiny', config='whisper_tiny_like', wseed=0, batch=4,
         frames=(37, 200), fseed=31, beam=5),
    dict(case='whisperenc_tiny_odd', config='whisper_tiny_like', wseed=1, batch=3,
         frames=(20, 151), fseed=32, beam=4),
    dict(case='whisperenc_largev3_2blocks', config='whisper_largev3_2blocks',
         wseed=0, batch=2, frames=(121, 180), fseed=33, beam=10),
]


def build_reference_encoder(configs, sd):
    _ref_harness.install()
    from wenet.models.transformer.ctc import CTC
    from wenet.models.transformer.encoder import TransformerEncoder
    enc = TransformerEncoder(configs['input_dim'], global_cmvn=None,
                             **configs['encoder_conf'])
    enc.load_state_dict({k[len('encoder.'):]: v for k, v in sd.items()
                         if k.startswith('encoder.')}, strict=True)
    ctc = CTC(configs['output_dim'], configs['encoder_conf']['output_size'])
    ctc.load_state_dict({k[len('ctc.'):]: v for k, v in sd.items()
                         if k.startswith('ctc.')}, strict=True)
    return enc.eval(), ctc.eval()


def main():
    _ref_harness.install()
    from wenet.models.transformer import search as ref_search
    from wenet_amd import synthetic as S
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    outdir = os.path.join(ROOT, 'tests', 'golden')
    for c in CASES:
        configs = S.make_configs(c['config'])
        sd = S.make_state_dict(configs, c['wseed'])
        enc, ctc = build_reference_encoder(configs, sd)
        feats, lens = S.make_features(c['batch'], c['frames'], seed=c['fseed'],
                                      feat_dim=configs['input_dim'])
        with torch.no_grad():
            out, mask = enc(feats, lens)
            enc_lens = mask.squeeze(1).sum(1)
            logp = ctc.log_softmax(out)
            greedy = ref_search.ctc_greedy_search(logp, enc_lens)
            prefix = ref_search.ctc_prefix_beam_search(logp, enc_lens, c['beam'])
        k = min(16, logp.size(-1))
        topv, topi = logp.topk(k, dim=-1)
        meta = dict(c)
        meta['greedy'] = [r.tokens for r in greedy]
        meta['prefix'] = [
            dict(nbest=[list(map(int, h)) for h in r.nbest],
                 nbest_scores=[float(s) for s in r.nbest_scores],
                 nbest_times=[list(map(int, t)) for t in r.nbest_times])
            for r in prefix]
        path = os.path.join(outdir, c['case'] + '.npz')
        np.savez_compressed(
            path, enc_out=out.numpy().astype(np.float32),
            enc_lens=enc_lens.numpy().astype(np.int32),
            ctc_topk_val=topv.numpy().astype(np.float32),
            ctc_topk_idx=topi.numpy().astype(np.int32),
            meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
        print(path, os.path.getsize(path) // 1024, 'KiB', enc_lens.tolist(),
              [len(g) for g in meta['greedy']])


if __name__ == '__main__':
    main()
