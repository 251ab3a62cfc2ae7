#!/usr/bin/env python3
"""Calibrate bench.py's `cpu_baseline` (kind "port": oracle/wenet_oracle.py) against
the REAL reference: both decode the bench batch (synthetic.make_bench_batch) with the
same weights, method, beam and torch thread count on THIS container's cores.  The GPU
box has no /root/reference, so the ratio measured here is what turns the port's
number on the GPU box into a reference-equivalent one.

    python tools/cpu_port_vs_reference.py [config2] > profiles/cpu_port_vs_reference.json
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, reps):
    fn()  # warm-up
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2], ts


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else 'config2'
    from oracle import _ref_harness, wenet_oracle as O
    from oracle.gen_golden import build_reference_model
    from wenet_amd import synthetic as S
    _ref_harness.install()
    wl = S.BENCH_WORKLOADS[workload]
    configs = S.make_configs(wl['config'])
    sd = S.make_state_dict(configs, 0)
    feats, lens = S.make_bench_batch(workload, 1)
    audio = float(sum(((int(t) - 1) * 160 + 400) / 16000.0 for t in lens.tolist()))
    nthr = os.cpu_count() or 1
    torch.set_num_threads(nthr)
    ref_model = build_reference_model(configs, sd)
    kw = dict(wl['kw'])
    method = wl['method']

    def run_ref():
        with torch.no_grad():
            return ref_model.decode([method], feats, lens, beam_size=S.BENCH_BEAM, **kw)

    def run_port():
        return O.decode(configs, sd, [method], feats, lens, beam_size=S.BENCH_BEAM, **kw)

    a, b = run_ref()[method], run_port()[method]
    same = sum(list(x.tokens) == list(y.tokens) for x, y in zip(a, b))
    t_ref, all_ref = timed(run_ref, 3)
    t_port, all_port = timed(run_port, 3)
    print(json.dumps({
        'workload': wl['text'], 'method': method, 'beam': S.BENCH_BEAM,
        'utterances': int(feats.size(0)), 'audio_seconds': round(audio, 1),
        'threads': nthr, 'host': 'build container (no GPU)',
        'reference_ASRModel_decode': {'seconds_median': round(t_ref, 3),
                                      'audio_s_per_s': round(audio / t_ref, 2),
                                      'runs': [round(x, 3) for x in all_ref]},
        'oracle_port_decode': {'seconds_median': round(t_port, 3),
                               'audio_s_per_s': round(audio / t_port, 2),
                               'runs': [round(x, 3) for x in all_port]},
        'port_over_reference_speed': round(t_ref / t_port, 3),
        'identical_1best': f'{same}/{len(a)}',
        'note': 'port = oracle/wenet_oracle.py (torch-CPU fp32 restatement + Python '
                'prefix beam); reference = /root/reference wenet ASRModel.decode imported '
                'unmodified through oracle/_ref_harness.py; same tensors, same thread '
                'count, median of 3 after 1 warm-up',
    }, indent=1))


if __name__ == '__main__':
    main()
