#!/usr/bin/env python3
"""One short GPU visit for the kernel forms prepared without a GPU:
each key against the default on the config-2 bench batch -- encoder output bit for bit, a
crude time of the encoder -- the CTC top-k form through decode(), and the asm transpose reads
of the bf16 attention on a small Whisper-like model.  Prints one line per form; progress is
flushed line by line so that a cut-off visit still tells what it reached."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
T0 = time.time()


def say(*a):
    print(f'[{time.time() - T0:6.1f}s]', *a, flush=True)


import torch  # noqa: E402
say('torch imported')
from gpu_util import cached_model  # noqa: E402
from wenet_amd import _lib, synthetic as S  # noqa: E402

L = _lib.lib()


def tune(k, v):
    _lib.check(L.wn_tune_set(k.encode(), v), 'tune')


def timed(f, n=6):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


wl = S.BENCH_WORKLOADS['config2']
configs, sd, model = cached_model(wl['config'], 0)
feats, lens = S.make_bench_batch('config2', 1)
feats = feats.cuda()
say('model + batch ready', tuple(feats.shape))


def enc():
    e, _ = model._forward_encoder(feats, lens)
    return e


base = enc().clone()
t_base = timed(enc)
say(f'default encoder {t_base:.3f} ms')
for key, val, dflt in (('x6r_pro', 2, 1), ('attn_gload', 1, 0), ('dwconv_tiled', 1, 0)):
    try:
        tune(key, val)
        e = enc().clone()
        e2 = enc().clone()
        t = timed(enc)
        say(f'{key}={val}: bit-identical {torch.equal(e, base)} deterministic '
            f'{torch.equal(e, e2)} max|d| {(e - base).abs().max().item():.3e} '
            f'encoder {t:.3f} ms (default {t_base:.3f})')
    except Exception as ex:  # noqa: BLE001
        say(f'{key}={val}: FAILED {ex!r}')
    finally:
        tune(key, dflt)
t_base2 = timed(enc)
say(f'default encoder again {t_base2:.3f} ms')

kw = dict(beam_size=10)
m = 'ctc_prefix_beam_search'
r1 = model.decode([m], feats, lens, **kw)[m]
t1 = timed(lambda: model.decode([m], feats, lens, **kw))
try:
    tune('ctc_wave', 2)
    r2 = model.decode([m], feats, lens, **kw)[m]
    t2 = timed(lambda: model.decode([m], feats, lens, **kw))
    same = all(a.tokens == b.tokens and a.score == b.score and a.nbest == b.nbest and
               a.nbest_scores == b.nbest_scores and a.nbest_times == b.nbest_times
               for a, b in zip(r1, r2))
    say(f'ctc_wave=2: n-best lists, scores and times identical {same}; plain decode '
        f'{t2:.3f} ms (default {t1:.3f})')
except Exception as ex:  # noqa: BLE001
    say(f'ctc_wave=2: FAILED {ex!r}')
finally:
    tune('ctc_wave', 1)

# bf16 attention: asm transpose reads against the builtin ones and the register-staged kernel
try:
    cw, sdw, mw = cached_model('whisper_tiny_like', 0)
    fw, lw = S.make_features(3, (900, 2600), seed=6, feat_dim=cw['input_dim'])
    fw = fw.cuda()
    mw.set_compute_dtype('bf16')
    tune('attn_bf16_defer', 0)
    outs = {}
    for dma in (4, 5, 0):
        tune('attn_bf16_dma', dma)
        e, _ = mw._forward_encoder(fw, lw)
        outs[dma] = e.clone()
    e5b, _ = mw._forward_encoder(fw, lw)
    say(f'attn_bf16_dma=5: equals dma=4 {torch.equal(outs[5], outs[4])} equals dma=0 '
        f'{torch.equal(outs[5], outs[0])} max|d| {(outs[5] - outs[4]).abs().max().item():.3e}')
    tune('attn_bf16_defer', 80)
    for dma in (4, 5):
        tune('attn_bf16_dma', dma)
        t = timed(lambda: mw._forward_encoder(fw, lw), 4)
        say(f'attn_bf16_dma={dma}: small Whisper-like encoder {t:.3f} ms')
except Exception as ex:  # noqa: BLE001
    say(f'attn_bf16_dma=5: FAILED {ex!r}')
finally:
    tune('attn_bf16_dma', 4)
    tune('attn_bf16_defer', 80)
say('done')
