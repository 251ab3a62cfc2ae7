#!/usr/bin/env python3
"""Latency of a plain ASRModel.decode() at small batch sizes (a single transcribe(), a handful of
utterances): wall time per decode and the host time of the encoder's launch sequence alone
(wn_encode returns when everything is queued), to see where a decode turns launch-bound.

    python tools/latency_small_batch.py [config2|config3|config4] [B ...]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wenet_amd import synthetic as S  # noqa: E402
from wenet_amd.model import ASRModel  # noqa: E402

from wenet_amd import _lib  # noqa: E402
for kv in filter(None, os.environ.get('WN_TUNE', '').split(',')):     # WN_TUNE=key=value,...
    k, v = kv.split('=')
    _lib.check(_lib.lib().wn_tune_set(k.encode(), int(v)), 'tune')
wlname = sys.argv[1] if len(sys.argv) > 1 else 'config2'
sizes = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8, 16, 32]
wl = S.BENCH_WORKLOADS[wlname]
configs = S.make_configs(wl['config'])
model = ASRModel(configs, S.make_state_dict(configs, 0), device='cuda:0')
feats, lens = S.make_bench_batch(wlname, 1)
kw = dict(beam_size=S.BENCH_BEAM, **wl['kw'])
for B in sizes:
    fd = feats[:B, :int(lens[:B].max())].contiguous().cuda()
    ln = lens[:B]
    audio = float((ln.double() * 0.01).sum())
    for _ in range(5):
        model.decode([wl['method']], fd, ln, **kw)
    torch.cuda.synchronize()
    n = 40
    t0 = time.perf_counter()
    for _ in range(n):
        model.decode([wl['method']], fd, ln, **kw)
    torch.cuda.synchronize()
    dec = (time.perf_counter() - t0) / n
    # encoder alone: host time to queue it, and queue + sync
    chunk = kw.get('decoding_chunk_size', -1)
    tq = ts = 0.0
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model._forward_encoder(fd, ln, chunk, -1)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        tq += t1 - t0
        ts += t2 - t0
    print(f'{wlname} B={B:3d} ({audio:6.1f} s of audio): decode {dec * 1e3:7.3f} ms '
          f'({audio / dec:8.0f} audio-s/s); encoder alone: queued in {tq / n * 1e3:6.3f} ms, '
          f'done in {ts / n * 1e3:6.3f} ms', flush=True)
