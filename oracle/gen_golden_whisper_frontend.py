#!/usr/bin/env python3
"""tests/golden/whisperfb_*.npz: output of the REFERENCE's C++ frontend in its Whisper
configuration (oracle/_ref, built from /root/reference/runtime/core/frontend by
oracle/Makefile) on seeded waveforms, plus its Slaney filter bank and window.  Build
container only; the fixtures travel to the GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_fbank  # noqa: E402
from wenet_amd import synthetic as S  # noqa: E402


def main():
    out = os.path.join(ROOT, 'tests', 'golden')
    w80, win = ref_fbank.ref_slaney_filters(80)
    w128, _ = ref_fbank.ref_slaney_filters(128)
    np.savez_compressed(os.path.join(out, 'whisperfb_filters.npz'), w80=w80, w128=w128,
                        window=win)
    for name, n, seed, bins in [('a', 16000 * 3, 1, 80), ('b', 16000 * 2 + 37, 2, 128)]:
        wave = np.asarray(S.make_audio(n, seed=seed), dtype=np.float32)
        feat = ref_fbank.ref_whisper_fbank(wave, bins)
        np.savez_compressed(os.path.join(out, f'whisperfb_{name}.npz'), feat=feat,
                            n=np.int64(n), seed=np.int64(seed), bins=np.int64(bins))
        print(name, feat.shape, float(feat.min()), float(feat.max()))


if __name__ == '__main__':
    main()
