#!/usr/bin/env python3
"""Third short visit: attn_bf16_dma = 5 (transpose reads as inline asm, no compiler vmcnt(0)
in front of them) against the default 4 on the config-5 encoder (Whisper-large-v3 shape,
B = 16 x 3000 frames), bf16 and fp8 modes: bit-identity and interleaved time."""
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
from wenet_amd import _lib, synthetic as S  # noqa: E402
from wenet_amd.model import ASRModel  # noqa: E402

L = _lib.lib()


def tune(k, v):
    _lib.check(L.wn_tune_set(k.encode(), v), 'tune')


wl = S.BENCH_WORKLOADS['config5']
configs = S.make_configs(wl['config'])
sd = S.make_state_dict(configs, 0)
say('state dict ready')
model = ASRModel(configs, sd, device=torch.device('cuda', 0))
feats, lens = S.make_bench_batch('config5', 1)
feats = feats.cuda()
say('model + batch ready', tuple(feats.shape))


def enc():
    e, _ = model._forward_encoder(feats, lens)
    return e


def timed(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        enc()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for dt in ('bf16', 'fp8'):
    try:
        model.set_compute_dtype(dt)
        tune('attn_bf16_dma', 4)
        e4 = enc().clone()
        tune('attn_bf16_dma', 5)
        e5 = enc().clone()
        say(f'{dt}: dma=5 equals dma=4 bit for bit {torch.equal(e4, e5)} '
            f'max|d| {(e4 - e5).abs().max().item():.3e} finite {bool(torch.isfinite(e5).all())}')
        del e4, e5
        r = {4: [], 5: []}
        for _ in range(4):
            for dma in (4, 5):
                tune('attn_bf16_dma', dma)
                enc()
                r[dma].append(timed(3))
        for dma in (4, 5):
            say(f'{dt}: attn_bf16_dma={dma} encoder median {statistics.median(r[dma]):.3f} ms '
                f'(min {min(r[dma]):.3f}, max {max(r[dma]):.3f})')
    except Exception as ex:  # noqa: BLE001
        say(f'{dt}: FAILED {ex!r}')
    finally:
        tune('attn_bf16_dma', 4)
say('done')
