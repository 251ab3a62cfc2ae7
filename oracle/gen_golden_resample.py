#!/usr/bin/env python3
"""tests/golden/resample_*.npz: a SECOND, independent statement of what
`processor.resample` (wenet/dataset/processor.py:177-196 = torchaudio.transforms.Resample
with its defaults) computes, evaluated in fp64 -- the pin for `oracle.wenet_oracle.resample`
and `wn_resample` while torchaudio itself is unavailable (not vendored under /root/reference,
not installed; requirements.txt: torchaudio>=2.1.2).

Independent of the oracle's polyphase restatement in every step that can go wrong there (gcd
reduction of the kernel table, the (width, width + orig) zero padding, the strided windows,
the fp32 kernel table, the output-length rule): this file evaluates the DEFINITION the
published torchaudio kernel implements, sample by sample --

    y[m] = (B / orig) * sum_n x[n] * sinc(pi t) * cos^2(pi t / (2 W)),   |t| < W,
    t = (n / orig - m / new) * B,   B = 0.99 * min(orig, new),   W = 6,

with x = 0 outside the signal, orig / new reduced by their gcd, for m = 0 .. ceil(new * N /
orig) - 1 (torchaudio `_get_sinc_resample_kernel`: `t = (-i / new + idx / orig) * base`
clamped to [-W, W], window cos^2(t pi / W / 2), scale base / orig; `_apply_sinc_resample_kernel`:
target length ceil(new * length / orig)).  No tables, no padding arithmetic, fp64 throughout.

    python oracle/gen_golden_resample.py
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = [(44100, 16000), (48000, 16000), (8000, 16000), (22050, 16000), (16000, 8000),
         (11025, 16000), (32000, 16000)]


def resample_direct_fp64(x, orig_freq, new_freq, lpw=6, rolloff=0.99):
    x = np.asarray(x, dtype=np.float64)
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    n_out = -(-new * len(x) // orig)
    y = np.zeros(n_out, dtype=np.float64)
    half = lpw * orig / base          # support in input samples around m * orig / new
    for m in range(n_out):
        c = m * orig / new
        lo, hi = max(0, int(math.floor(c - half)) - 1), min(len(x) - 1, int(math.ceil(c + half)) + 1)
        if hi < lo:
            continue
        n = np.arange(lo, hi + 1, dtype=np.float64)
        t = (n / orig - m / new) * base
        keep = np.abs(t) < lpw
        t = t[keep]
        with np.errstate(invalid='ignore', divide='ignore'):
            s = np.where(t == 0.0, 1.0, np.sin(np.pi * t) / (np.pi * t))
        w = np.cos(np.pi * t / (2 * lpw)) ** 2
        y[m] = (base / orig) * np.dot(x[lo:hi + 1][keep], s * w)
    return y


def main():
    from wenet_amd import synthetic as S
    outdir = os.path.join(ROOT, 'tests', 'golden')
    # --from-torchaudio: the REAL third-party implementation the reference calls
    # (processor.resample -> torchaudio.transforms.Resample(orig, new)(wav), fp32) instead of
    # the fp64 definition; the files then carry source='torchaudio <version>' and
    # tests/test_oracle.py reports the pin as a reference run.  Not possible in this image
    # (torchaudio is not installed and there is no network): until then the resample row
    # stays "parity unpinned" (README.md).
    from_ta = '--from-torchaudio' in sys.argv
    if from_ta:
        import torch
        import torchaudio       # noqa: F401 -- fails loudly where it is absent
    for orig, new in CASES:
        n = int(0.2 * orig) + 37                    # odd length: exercises the ceil rule
        x = S.make_audio(n, seed=900 + orig // 100, sample_rate=orig)
        if from_ta:
            xt = torch.from_numpy(np.asarray(x, dtype=np.float32)).unsqueeze(0)
            y = torchaudio.transforms.Resample(orig_freq=orig, new_freq=new)(xt)[0].numpy()
            y = y.astype(np.float64)
            source = 'torchaudio ' + torchaudio.__version__
        else:
            y = resample_direct_fp64(x, orig, new)
            source = 'fp64 definition (parity unpinned)'
        path = os.path.join(outdir, f'resample_{orig}_{new}.npz')
        np.savez_compressed(path, x=x.astype(np.float32), y=y, orig=orig, new=new,
                            source=source)
        print(path, len(x), '->', len(y), source)


if __name__ == '__main__':
    main()
