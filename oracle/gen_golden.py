#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (imported
unmodified from /root/reference through oracle/_ref_harness.py) on seeded
synthetic models and inputs.  Runs only in the build container (the reference
tree does not exist on the GPU box); the fixtures it writes are committed.

    python oracle/gen_golden.py            # rewrites every fixture

Weights and inputs are NOT stored: they are regenerated bit-identically from
(config name, seed) by wenet_amd/synthetic.py.  Stored are the reference's
outputs at each stage boundary.
"""
import argparse
import copy
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import _ref_harness  # noqa: E402

CASES = [
    # name, config, weight seed, batch, frames, feat seed, beam, chunk, left,
    # ctc_weight, reverse_weight
    dict(case='tiny_causal_full', config='tiny_causal', wseed=0, batch=3,
         frames=(90, 140), fseed=5, beam=5, chunk=-1, left=-1, ctc_weight=0.5,
         reverse_weight=0.3),
    dict(case='tiny_causal_chunk4', config='tiny_causal', wseed=0, batch=3,
         frames=(90, 140), fseed=6, beam=4, chunk=4, left=-1, ctc_weight=0.3,
         reverse_weight=0.5),
    dict(case='tiny_causal_chunk4_left2', config='tiny_causal', wseed=1,
         batch=4, frames=(60, 200), fseed=7, beam=3, chunk=4, left=2,
         ctc_weight=0.0, reverse_weight=0.0),
    dict(case='tiny_sym_full', config='tiny_sym', wseed=0, batch=5,
         frames=(40, 170), fseed=8, beam=6, chunk=-1, left=-1, ctc_weight=0.5,
         reverse_weight=0.0),
    dict(case='aishell_full', config='aishell_u2pp', wseed=0, batch=3,
         frames=(250, 420), fseed=9, beam=10, chunk=-1, left=-1,
         ctc_weight=0.5, reverse_weight=0.3),
    dict(case='aishell_chunk16', config='aishell_u2pp', wseed=0, batch=2,
         frames=(300, 380), fseed=10, beam=10, chunk=16, left=-1,
         ctc_weight=0.5, reverse_weight=0.3),
    dict(case='librispeech_full', config='librispeech_bidecoder_large',
         wseed=0, batch=2, frames=(260, 330), fseed=11, beam=10, chunk=-1,
         left=-1, ctc_weight=0.5, reverse_weight=0.3),
    dict(case='tiny_bn_full', config='tiny_bn', wseed=0, batch=4,
         frames=(50, 180), fseed=13, beam=5, chunk=-1, left=-1, ctc_weight=0.5,
         reverse_weight=0.0),
    dict(case='aishell_conformer_full', config='aishell_conformer', wseed=0,
         batch=2, frames=(200, 300), fseed=14, beam=10, chunk=-1, left=-1,
         ctc_weight=0.5, reverse_weight=0.0),
    # model_conf.apply_non_blank_embedding (train_u2++_lite_conformer.yaml): rescoring on the
    # non-blank frames.  One utterance / equal lengths: no padded frame enters the filter
    dict(case='tiny_lite_one', config='tiny_lite', wseed=0, batch=1,
         frames=(150, 150), fseed=21, beam=5, chunk=-1, left=-1, ctc_weight=0.5,
         reverse_weight=0.3),
    dict(case='tiny_lite_equal', config='tiny_lite', wseed=1, batch=4,
         frames=(131, 131), fseed=22, beam=4, chunk=-1, left=-1, ctc_weight=0.3,
         reverse_weight=0.0),
    dict(case='aishell_lite_one', config='aishell_u2pp_lite', wseed=0, batch=1,
         frames=(330, 330), fseed=23, beam=10, chunk=-1, left=-1, ctc_weight=0.5,
         reverse_weight=0.3),
    # ... and a RAGGED batch: the reference's filter_blank_embedding takes the arg-max over all
    # maxlen frames, the padded ones of the shorter utterances included (asr_model.py:153-180),
    # and attention_rescoring slices the result with the unfiltered lengths.  The file prefix
    # keeps it out of the generic case list (tests/golden_util.py): the accelerated path has
    # no padded frames (tests/test_gpu_parity.py)
    dict(case='raggedlite_tiny', config='tiny_lite', wseed=0, batch=5,
         frames=(60, 190), fseed=4242, beam=5, chunk=-1, left=-1, ctc_weight=0.5,
         reverse_weight=0.3),
    # (weights under which padded frames DO come out non-blank: the reference keeps
    # [19, 20, 29, 32, 31] rows where only [19, 20, 25, 12, 8] belong to the utterances)
    dict(case='raggedlite_tiny_padded', config='tiny_lite', wseed=4, batch=5,
         frames=(60, 190), fseed=4242, beam=5, chunk=-1, left=-1, ctc_weight=0.5,
         reverse_weight=0.3),
    dict(case='wenetspeech_chunk16', config='wenetspeech_u2pp', wseed=0,
         batch=2, frames=(260, 330), fseed=12, beam=10, chunk=16, left=-1,
         ctc_weight=0.5, reverse_weight=0.3),
]


def build_reference_model(configs, sd):
    """Reference ASRModel holding the synthetic weights."""
    _ref_harness.install()
    from wenet.models.transformer.cmvn import GlobalCMVN
    from wenet.utils.init_model import init_model
    rc = copy.deepcopy(configs)
    rc['cmvn'] = None
    model, _ = init_model(argparse.Namespace(), rc)
    model.encoder.global_cmvn = GlobalCMVN(sd['encoder.global_cmvn.mean'],
                                           sd['encoder.global_cmvn.istd'])
    model.load_state_dict(sd, strict=True)
    model.eval()
    return model


def run_case(c, outdir):
    from wenet_amd import synthetic as S
    configs = S.make_configs(c['config'])
    sd = S.make_state_dict(configs, c['wseed'])
    model = build_reference_model(configs, sd)
    feats, lens = S.make_features(c['batch'], c['frames'], seed=c['fseed'])
    methods = ['ctc_greedy_search', 'ctc_prefix_beam_search',
               'attention_rescoring']
    with torch.no_grad():
        enc, mask = model._forward_encoder(feats, lens, c['chunk'], c['left'])
        enc_lens = mask.squeeze(1).sum(1)
        logp = model.ctc_logprobs(enc)
        res = model.decode(methods, feats, lens, beam_size=c['beam'],
                           decoding_chunk_size=c['chunk'],
                           num_decoding_left_chunks=c['left'],
                           ctc_weight=c['ctc_weight'],
                           reverse_weight=c['reverse_weight'])
    k = min(16, logp.size(-1))
    topv, topi = logp.topk(k, dim=-1)
    out = dict(
        enc_out=enc.numpy().astype(np.float32),
        enc_lens=enc_lens.numpy().astype(np.int32),
        ctc_topk_val=topv.numpy().astype(np.float32),
        ctc_topk_idx=topi.numpy().astype(np.int32),
    )
    if logp.numel() <= 200000:
        out['ctc_logp'] = logp.numpy().astype(np.float32)
    meta = dict(c)
    if getattr(model, 'apply_non_blank_embedding', False):
        # what attention_rescoring attended to: the reference's own filter_blank_embedding
        with torch.no_grad():
            sel, smask = model.filter_blank_embedding(logp, enc)
        out['nonblank_out'] = sel.numpy().astype(np.float32)
        meta['nonblank_kept'] = [int(x) for x in smask.squeeze(1).sum(1).tolist()]
    meta['greedy'] = [r.tokens for r in res['ctc_greedy_search']]
    meta['prefix'] = [
        dict(nbest=[list(map(int, h)) for h in r.nbest],
             nbest_scores=[float(s) for s in r.nbest_scores],
             nbest_times=[list(map(int, t)) for t in r.nbest_times])
        for r in res['ctc_prefix_beam_search']
    ]
    meta['rescoring'] = [
        dict(tokens=list(map(int, r.tokens)), score=float(r.score),
             confidence=float(r.confidence),
             tokens_confidence=[float(x) for x in r.tokens_confidence],
             times=list(map(int, r.times)))
        for r in res['attention_rescoring']
    ]
    out['meta'] = np.frombuffer(json.dumps(meta).encode('utf8'),
                                dtype=np.uint8)
    path = os.path.join(outdir, f"{c['case']}.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, 'KiB', 'enc_lens',
          enc_lens.tolist(), 'greedy lens',
          [len(g) for g in meta['greedy']])


def main():
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    outdir = os.path.join(ROOT, 'tests', 'golden')
    os.makedirs(outdir, exist_ok=True)
    only = set(sys.argv[1:])
    for c in CASES:
        if only and c['case'] not in only:
            continue
        run_case(c, outdir)


if __name__ == '__main__':
    main()
