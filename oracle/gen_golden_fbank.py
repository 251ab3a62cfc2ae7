#!/usr/bin/env python3
"""Generate tests/golden/fbank_*.npz with the REFERENCE's own C++ fbank
(oracle/_ref/libref_fbank.so, built from /root/reference by oracle/Makefile) on
seeded synthetic audio.  Only the outputs are stored; the waveforms are
regenerated bit-identically by wenet_amd/synthetic.py::make_audio.

    make -C oracle && python oracle/gen_golden_fbank.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_fbank  # noqa: E402
from wenet_amd import synthetic as S  # noqa: E402

CASES = [dict(case='fbank_a', samples=16000, seed=0),
         dict(case='fbank_b', samples=5000, seed=1),
         dict(case='fbank_short', samples=399, seed=2),
         dict(case='fbank_c', samples=24321, seed=3),
         dict(case='fbank_min', samples=400, seed=4)]


def main():
    outdir = os.path.join(ROOT, 'tests', 'golden')
    for c in CASES:
        w = S.make_audio(c['samples'], seed=c['seed'])
        f = ref_fbank.ref_fbank(w)
        meta = dict(c, frames=int(f.shape[0]),
                    wave_sum=float(np.sum(w.astype(np.float64))))
        path = os.path.join(outdir, c['case'] + '.npz')
        np.savez_compressed(path, feats=f,
                            meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
        print(path, f.shape, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
