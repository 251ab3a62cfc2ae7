#!/usr/bin/env python3
"""Throughput of the REAL data path: wav files on local disk -> `python -m wenet_amd.bin.recognize`
-> result text (`recognize.py` data side, wenet/bin/recognize.py:
282-311, wenet/dataset/processor.py:125-153,526-577), against what bench.py measures with the
features already in HBM.

    python tools/bench_datapath.py [--hours 1.0] [--workers 1,4,8,16] [--out gpurun_out/x.json]

Writes ~`hours` of synthetic 16 kHz PCM16 wavs (8-12 s each, the bench batch's length
distribution) to --dir, a raw list and a shard (tar) list, a BASELINE configs[1] model directory
(AIShell u2++ conformer, random-init weights), then runs the CLI in-process for every
(data_type, num_workers) with --report_rtf and prints / stores the reports.
"""
import argparse
import io
import json
import os
import sys
import tarfile
import time
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_corpus(d, hours, seed=1234):
    from wenet_amd import synthetic as S
    rng = np.random.Generator(np.random.PCG64([seed, 3]))
    os.makedirs(os.path.join(d, 'wav'), exist_ok=True)
    total, i, entries = 0.0, 0, []
    base = [S.make_audio(12 * 16000, seed=seed + k) for k in range(16)]   # 16 voices, reused
    while total < hours * 3600.0:
        n = int(rng.integers(8 * 16000, 12 * 16000 + 1))
        x = base[i % len(base)][:n] * float(rng.uniform(0.5, 1.0))
        x = np.roll(x, int(rng.integers(0, n)))
        pcm = np.clip(np.round(x * 32768.0), -32768, 32767).astype('<i2')
        path = os.path.join(d, 'wav', f'utt{i:06d}.wav')
        with wave.open(path, 'wb') as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
            w.writeframes(pcm.tobytes())
        entries.append((f'utt{i:06d}', path))
        total += n / 16000.0
        i += 1
    raw = os.path.join(d, 'raw.list')
    with open(raw, 'w') as f:
        for k, p in entries:
            f.write(json.dumps(dict(key=k, wav=p, txt='')) + '\n')
    # shards of 200 utterances: <key>.wav + <key>.txt members (datapipes.py:365-427)
    shard_list = os.path.join(d, 'shard.list')
    with open(shard_list, 'w') as fl:
        for s0 in range(0, len(entries), 200):
            tp = os.path.join(d, f'shard_{s0 // 200:04d}.tar')
            with tarfile.open(tp, 'w') as tar:
                for k, p in entries[s0:s0 + 200]:
                    tar.add(p, arcname=k + '.wav')
                    ti = tarfile.TarInfo(k + '.txt')
                    ti.size = 0
                    tar.addfile(ti, io.BytesIO(b''))
            fl.write(tp + '\n')
    return raw, shard_list, total, len(entries)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--hours', type=float, default=1.0)
    ap.add_argument('--dir', default='/tmp/wn_datapath')
    ap.add_argument('--workers', default='1,4,8,16')
    ap.add_argument('--types', default='raw,shard')
    ap.add_argument('--batch_size', type=int, default=32)
    ap.add_argument('--streams', type=int, default=2)
    ap.add_argument('--out', default='')
    args = ap.parse_args()
    from wenet_amd import synthetic as S
    from wenet_amd.bin import recognize as R
    t0 = time.perf_counter()
    raw, shard, audio_s, n = write_corpus(args.dir, args.hours)
    mdir = S.write_model_dir(os.path.join(args.dir, 'model'), 'aishell_u2pp', 0)
    print(f'corpus: {n} wavs, {audio_s / 3600:.2f} h, written in '
          f'{time.perf_counter() - t0:.1f} s', flush=True)
    reports = []
    for dt in args.types.split(','):
        for nw in [int(x) for x in args.workers.split(',')]:
            rep = os.path.join(args.dir, f'rtf_{dt}_{nw}.json')
            argv = ['--config', os.path.join(mdir, 'train.yaml'), '--checkpoint',
                    os.path.join(mdir, 'final.pt'), '--test_data', raw if dt == 'raw' else shard,
                    '--data_type', dt, '--result_dir', os.path.join(args.dir, f'out_{dt}_{nw}'),
                    '--modes', 'ctc_prefix_beam_search', '--batch_size', str(args.batch_size),
                    '--beam_size', '10', '--num_workers', str(nw), '--streams',
                    str(args.streams), '--report_rtf', rep, '--gpu', '0']
            import logging
            logging.disable(logging.INFO)
            R.main(argv)
            logging.disable(logging.NOTSET)
            r = json.load(open(rep))
            reports.append(r)
            print(json.dumps(dict(data_type=dt, num_workers=nw,
                                  audio_s_per_s=r['audio_seconds_per_second'],
                                  wall_s=r['wall_seconds'], main=r['main_thread_seconds'])),
                  flush=True)
    if args.out:
        with open(args.out, 'w') as f:
            json.dump(dict(corpus_hours=round(audio_s / 3600, 3), utterances=n,
                           runs=reports), f, indent=1)


if __name__ == '__main__':
    main()
