"""Encoder-only loop for kernel profiling / ablation timing (no search, no host results: safe
for WN_ABLATION variants whose outputs are wrong by design):

    rocprofv3 --kernel-trace --stats -d DIR -o prof -- python tools/prof_encoder.py \
        [--workload config2] [--steps 6] [--tune key=value,...]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.getcwd())
from wenet_amd import _lib, synthetic as S            # noqa: E402
from wenet_amd.model import ASRModel                  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--workload', default='config2')
ap.add_argument('--steps', type=int, default=6)
ap.add_argument('--tune', default='')
args = ap.parse_args()
wl = S.BENCH_WORKLOADS[args.workload]
configs = S.make_configs(wl['config'])
model = ASRModel(configs, S.make_state_dict(configs, 0), device='cuda:0')
for kv in filter(None, args.tune.split(',')):
    k, v = kv.split('=')
    _lib.check(_lib.lib().wn_tune_set(k.encode(), int(v)), 'tune')
feats, lens = S.make_bench_batch(args.workload, 1)
fd = feats.cuda()
kw = wl['kw']
chunk = kw.get('decoding_chunk_size', -1)
left = kw.get('num_decoding_left_chunks', -1)
for _ in range(2):
    model._forward_encoder(fd, lens, chunk, left)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    model._forward_encoder(fd, lens, chunk, left)
torch.cuda.synchronize()
print('encoder ms', (time.perf_counter() - t0) / args.steps * 1e3)
