#!/usr/bin/env python3
"""Latency / throughput of BATCHED cache streaming (SURVEY.md section 8 row f2,
`ASRModel.forward_encoder_chunk_batch` = wn_encode_chunk_batch; the reference's batched
formulation: wenet/bin/export_onnx_gpu.py:83-232; published context for the reference's own
GPU streaming server: runtime/gpu/README.md:140-186).

N concurrent sessions of the WenetSpeech u2++ conformer (12L / 8 heads / 512d, causal,
chunk-trained: BASELINE.json configs[3]'s model), decoding_chunk_size 16 (= 0.64 s of audio per
step and session), num_decoding_left_chunks L (required_cache_size = 16 L; -1 = all history
up to --max-cache frames).  One STEP = every session advances by one chunk: the encoder chunk
forward for all N sessions in one call + the CTC head on the N x 16 new frames (log-softmax +
top-10, what a streaming prefix beam consumes).  Reported per N: median / p99 step latency in
steady state (caches full) and audio-seconds per second = N x 0.64 / step.

    python tools/bench_streaming.py [--sessions 16,32,64] [--left 4] [--steps 200]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sessions', default='1,16,32,64')
    ap.add_argument('--left', type=int, default=4, help='num_decoding_left_chunks (-1: all)')
    ap.add_argument('--chunk', type=int, default=16)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--max-cache', type=int, default=512)
    ap.add_argument('--config', default='wenetspeech_u2pp')
    ap.add_argument('--out', default='')
    args = ap.parse_args()
    from wenet_amd import synthetic as S
    from wenet_amd.model import ASRModel
    dev = torch.device('cuda', 0)
    configs = S.make_configs(args.config)
    model = ASRModel(configs, S.make_state_dict(configs, 0), device=dev)
    chunk = args.chunk
    window = (chunk - 1) * 4 + 7          # feature frames per step (encoder.py:337-352)
    req = chunk * args.left if args.left >= 0 else -1
    rows = []
    for n in [int(x) for x in args.sessions.split(',')]:
        feats, _ = S.make_features(n, window, seed=5, feat_dim=configs['input_dim'])
        xs = feats.to(dev)
        att = [None] * n
        cnn = [None] * n
        offsets = [0] * n
        lat = []
        for it in range(args.warmup + args.steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ys, att, cnn = model.forward_encoder_chunk_batch(xs, offsets, req, att, cnn)
            logp = model.ctc_logprobs(ys)                  # (n, chunk, V) log-softmax
            top = logp.topk(10, dim=-1)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            offsets = [o + chunk for o in offsets]
            if args.left < 0 and att[0].size(2) > args.max_cache:
                att = [a[:, :, -args.max_cache:].contiguous() for a in att]
            if it >= args.warmup:
                lat.append(dt)
        lat = np.asarray(lat) * 1e3
        audio = n * chunk * 0.04
        row = dict(sessions=n, chunk=chunk, left_chunks=args.left,
                   cache_frames=int(att[0].size(2)),
                   step_ms_median=round(float(np.median(lat)), 3),
                   step_ms_p99=round(float(np.percentile(lat, 99)), 3),
                   audio_s_per_s=round(audio / (float(np.median(lat)) * 1e-3), 1),
                   audio_s_per_step=round(audio, 2))
        rows.append(row)
        print(json.dumps(row), flush=True)
        del top
    if args.out:
        with open(args.out, 'w') as f:
            json.dump(dict(model=args.config, note='encoder chunk forward for all sessions + '
                           'CTC head (log-softmax, top-10) per step, host-synchronous; one '
                           'MI355X', rows=rows), f, indent=1)


if __name__ == '__main__':
    main()
