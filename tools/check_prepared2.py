#!/usr/bin/env python3
"""Second short visit for the prepared kernel forms: (A) interleaved
timing of the config-2 encoder / plain decode per key, (B) bit-identity of "all prepared forms
on" against the shipped kernels over the shapes the GPU suite uses (both d_model = 256 and 512
models, chunk masks, utterances of a few frames, small vocabularies, wide beams)."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
T0 = time.time()


def say(*a):
    print(f'[{time.time() - T0:6.1f}s]', *a, flush=True)


import torch  # noqa: E402
from gpu_util import cached_model  # noqa: E402
from wenet_amd import _lib, synthetic as S  # noqa: E402

L = _lib.lib()
say('imports done')
ENC_KEYS = (('x6r_pro', 2, 1), ('attn_gload', 1, 0), ('dwconv_tiled', 1, 0))
ALL_KEYS = ENC_KEYS + (('ctc_wave', 2, 1), )


def tune(k, v):
    _lib.check(L.wn_tune_set(k.encode(), v), 'tune')


def set_keys(keys, on):
    for k, v, d in keys:
        tune(k, v if on else d)


def timed(f, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


# ---- (A) timing -------------------------------------------------------------------------
wl = S.BENCH_WORKLOADS['config2']
configs, sd, model = cached_model(wl['config'], 0)
feats, lens = S.make_bench_batch('config2', 1)
feats = feats.cuda()


def enc():
    model._forward_encoder(feats, lens)


def dec():
    model.decode(['ctc_prefix_beam_search'], feats, lens, beam_size=10)


settings = [('default', ())] + [(f'{k}={v}', ((k, v, d), )) for k, v, d in ENC_KEYS] + \
    [('all three', ENC_KEYS)]
res = {n: [] for n, _ in settings}
for f in (enc, enc):
    f()
for _ in range(5):
    for name, keys in settings:
        set_keys(keys, True)
        enc()
        res[name].append(timed(enc, 12))
        set_keys(keys, False)
for name, _ in settings:
    say(f'encoder {name:16s} median {statistics.median(res[name]):.3f} ms  '
        f'(min {min(res[name]):.3f}, max {max(res[name]):.3f})')
dres = {'default': [], 'ctc_wave=2': [], 'all four': []}
for _ in range(4):
    dres['default'].append(timed(dec, 8))
    tune('ctc_wave', 2)
    dres['ctc_wave=2'].append(timed(dec, 8))
    set_keys(ENC_KEYS, True)
    dres['all four'].append(timed(dec, 8))
    set_keys(ALL_KEYS, False)
for name, v in dres.items():
    say(f'plain decode {name:12s} median {statistics.median(v):.3f} ms (min {min(v):.3f})')

# ---- (B) bit-identity sweep -----------------------------------------------------------
CASES = [
    ('aishell_u2pp', 4, (400, 700), -1, -1, 10), ('aishell_u2pp', 3, (7, 90), 16, -1, 10),
    ('aishell_u2pp', 5, (1, 300), 4, 2, 4), ('aishell_u2pp', 2, (1100, 1300), -1, -1, 16),
    ('wenetspeech_u2pp', 3, (300, 500), 16, -1, 10), ('wenetspeech_u2pp', 2, (600, 900), -1, -1, 10),
    ('wenetspeech_u2pp', 4, (3, 200), 8, 1, 5),
    ('librispeech_bidecoder_large', 3, (200, 600), -1, -1, 10),
    ('tiny_causal', 6, (30, 260), -1, -1, 10), ('tiny_causal', 4, (30, 200), 8, 1, 16),
    ('tiny_sym', 5, (7, 180), -1, -1, 3),
]
METHODS = ['ctc_greedy_search', 'ctc_prefix_beam_search', 'attention_rescoring']
bad = 0
for ci, (cfg, B, frames, chunk, left, beam) in enumerate(CASES):
    try:
        c, s_, m = cached_model(cfg, 0)
        f, ln = S.make_features(B, frames, seed=300 + ci, feat_dim=c.get('input_dim', 80))
        f = f.cuda()
        rw = 0.3 if c.get('decoder') == 'bitransformer' else 0.0
        set_keys(ALL_KEYS, False)
        e0, _ = m._forward_encoder(f, ln, chunk, left)
        e0 = e0.clone()
        d0 = m.decode(METHODS, f, ln, beam_size=beam, ctc_weight=0.5, reverse_weight=rw,
                      decoding_chunk_size=chunk, num_decoding_left_chunks=left)
        set_keys(ALL_KEYS, True)
        e1, _ = m._forward_encoder(f, ln, chunk, left)
        e1 = e1.clone()
        d1 = m.decode(METHODS, f, ln, beam_size=beam, ctc_weight=0.5, reverse_weight=rw,
                      decoding_chunk_size=chunk, num_decoding_left_chunks=left)
        same_enc = torch.equal(e0, e1)
        same_dec = True
        for mm in METHODS:
            for a, b in zip(d0[mm], d1[mm]):
                same_dec &= a.tokens == b.tokens and a.score == b.score
        for a, b in zip(d0['ctc_prefix_beam_search'], d1['ctc_prefix_beam_search']):
            same_dec &= (a.nbest == b.nbest and a.nbest_scores == b.nbest_scores and
                         a.nbest_times == b.nbest_times)
        ok = same_enc and same_dec and bool(torch.isfinite(e1).all())
        bad += not ok
        say(f'{cfg} B={B} frames={frames} chunk={chunk} left={left} beam={beam}: encoder '
            f'identical {same_enc}, decode results identical {same_dec}')
    except Exception as ex:  # noqa: BLE001
        bad += 1
        say(f'{cfg} B={B} frames={frames}: FAILED {ex!r}')
    finally:
        set_keys(ALL_KEYS, False)
say(f'sweep done: {len(CASES) - bad} / {len(CASES)} cases identical')
