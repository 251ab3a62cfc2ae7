"""Helpers to load the committed reference fixtures (tests/golden/*.npz,
written by oracle/gen_golden.py from the real reference)."""
import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def case_names(prefix=''):
    return sorted(
        os.path.basename(p)[:-4]
        for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + '*.npz'))
        if not os.path.basename(p).startswith(('fbank', 'whisper', 'stream_', 'attn_', 'ctx', 'chunk_',
                                                 'cmvn_', 'bench_', 'resample_', 'raggedlite_')))


def load_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    meta = json.loads(bytes(z['meta']).decode('utf8'))
    arrays = {k: z[k] for k in z.files if k != 'meta'}
    return meta, arrays


def build_inputs(meta):
    """Regenerate (configs, state_dict, feats, lens) of a golden case."""
    from wenet_amd import synthetic as S
    configs = S.make_configs(meta['config'])
    sd = S.make_state_dict(configs, meta['wseed'])
    feats, lens = S.make_features(meta['batch'], tuple(meta['frames']),
                                  seed=meta['fseed'],
                                  feat_dim=configs['input_dim'])
    return configs, sd, feats, lens


def whisper_case_names():
    """Goldens of the Whisper-style TransformerEncoder (+ CTC head)."""
    return sorted(os.path.basename(p)[:-4]
                  for p in glob.glob(os.path.join(GOLDEN_DIR, 'whisperenc_*.npz')))


def stream_case_names():
    """Goldens of the reference's cache-based streaming path (simulate_streaming)."""
    return sorted(os.path.basename(p)[:-4]
                  for p in glob.glob(os.path.join(GOLDEN_DIR, 'stream_*.npz')))


def attention_case_names():
    """Goldens of the reference's `attention` (autoregressive) decode mode."""
    return sorted(os.path.basename(p)[:-4]
                  for p in glob.glob(os.path.join(GOLDEN_DIR, 'attn_*.npz')))


def context_case_names():
    """Goldens of context-biased decoding through a model (ContextGraph)."""
    return sorted(os.path.basename(p)[:-4]
                  for p in glob.glob(os.path.join(GOLDEN_DIR, 'ctx_*.npz')))


def context_search_case_names():
    """Goldens of the context-biased ctc_prefix_beam_search on seeded log-probs."""
    return sorted(os.path.basename(p)[:-4]
                  for p in glob.glob(os.path.join(GOLDEN_DIR, 'ctxsearch_*.npz')))


def chunk_case_names():
    """Goldens of the reference's forward_chunk API (outputs + caches)."""
    return sorted(os.path.basename(p)[:-4]
                  for p in glob.glob(os.path.join(GOLDEN_DIR, 'chunk_*.npz')))


def chunk_windows(n_frames, chunk):
    """Feature windows (start, end) of forward_chunk_by_chunk (encoder.py:337-352)
    for Conv2dSubsampling4: rate 4, right context 6."""
    context, stride = 7, 4 * chunk
    window = (chunk - 1) * 4 + context
    return [(cur, min(cur + window, n_frames))
            for cur in range(0, n_frames - context + 1, stride)]
