"""CPU oracle for the WeNet Conformer-ASR inference hot path.

TEST INFRASTRUCTURE ONLY.  This file is a *restatement* of the reference
algorithm (wenet-e2e/wenet, Python path) written as plain functions over a
``state_dict`` (name -> fp32 tensor) and the parsed ``train.yaml`` dict.  It is
imported only by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- never by ``wenet_amd/`` (the product
path), which fails loudly when its HIP library is missing.

Pinning (see tests/test_oracle.py and tests/golden/):
  * model + search: checked against the reference's own ``ASRModel.decode`` /
    ``search.py`` imported unmodified from /root/reference (oracle/_ref_harness),
    at the tiny fixture sizes committed under tests/golden/ and, when the
    reference tree is present, at the full BASELINE configs;
  * prefix beam search: additionally against the known-answer table of
    runtime/core/test/ctc_prefix_beam_search_test.cc:29-72;
  * fbank: against the reference's C++ frontend (runtime/core/frontend/fbank.h)
    compiled from where it lies into oracle/_ref/ (oracle/Makefile), golden rows
    committed in tests/golden/fbank_*.npz.  torchaudio.compliance.kaldi (the
    Python path's third-party fbank, unpinned `torchaudio>=2.1.2`) is absent, so
    the Python-side fbank parity is anchored on that C++ restatement;
  * context biasing, forward_chunk caches, attention mode, Whisper encoder:
    goldens generated from the unmodified reference (oracle/gen_golden_*.py);
  * `resample` ONLY: PARITY UNPINNED -- torchaudio's Resample is third party,
    absent and un-vendored, and the reference holds no test for it; restated
    from the published algorithm and checked through properties.

All arithmetic is fp32 like the reference default (wenet/bin/recognize.py:250);
the prefix-beam bookkeeping uses Python floats (fp64) exactly like search.py.
Citations are path:line under /root/reference/.
"""
import math
from collections import defaultdict, deque
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F
import contextlib

_TORCH_F = F


class _Bf16OperandFunctional:
    """torch.nn.functional with the operand rounding of the product's
    WN_PREC_BF16 mode (include/wenet_amd.h, `recognize.py --dtype bf16`,
    wenet/bin/recognize.py:250-255,278-280): every contraction the product runs
    on its GEMM kernels -- nn.Linear, kernel-1 / subsampling Conv1d, the
    d -> d Conv2d of Conv2dSubsampling4 -- sees both operands rounded to bf16
    (round to nearest even) and accumulates in fp32.  Everything else stays
    fp32: the 1 -> d Conv2d and the depthwise Conv1d (not GEMMs in the product),
    softmax, LayerNorm, and `linear_pos` (projected once at model creation, in
    fp32).  The attention matmuls round their operands too (`_mm` below): q (+
    pos_bias_u / v), k, the projected position rows, the probabilities and v --
    what torch.matmul does under autocast.  The product rounds the UN-normalised
    online-softmax probabilities, the oracle the normalised ones: same relative
    rounding, not bit-identical."""

    def __init__(self, exempt_weights=(), mx_weights=()):
        self._exempt = {w.data_ptr() for w in exempt_weights}
        self._mx = {w.data_ptr() for w in mx_weights}

    def __getattr__(self, name):
        return getattr(_TORCH_F, name)

    @staticmethod
    def _r(t):
        return t.to(torch.bfloat16).to(torch.float32)

    def linear(self, x, w, b=None):
        if w.data_ptr() in self._exempt:
            return _TORCH_F.linear(x, w, b)
        if w.data_ptr() in self._mx:   # WN_PREC_FP8: MXFP8 operands for the FFN GEMMs
            return _TORCH_F.linear(mx_round(x), mx_round(w), b)
        return _TORCH_F.linear(self._r(x), self._r(w), b)

    def conv1d(self, x, w, b=None, **kw):
        if kw.get('groups', 1) != 1:
            return _TORCH_F.conv1d(x, w, b, **kw)
        return _TORCH_F.conv1d(self._r(x), self._r(w), b, **kw)

    def conv2d(self, x, w, b=None, **kw):
        if w.size(1) == 1:
            return _TORCH_F.conv2d(x, w, b, **kw)
        return _TORCH_F.conv2d(self._r(x), self._r(w), b, **kw)


_MM_ROUND = False  # bf16_operands(): attention matmuls round their operands


def _mm(a, b):
    """torch.matmul of the attention score / context products."""
    if _MM_ROUND:
        a = a.to(torch.bfloat16).to(torch.float32)
        b = b.to(torch.bfloat16).to(torch.float32)
    return torch.matmul(a, b)


def mx_quantize(x: torch.Tensor):
    """OCP MXFP8 quantisation along the last dim (a multiple of 32): e4m3 elements
    (torch.float8_e4m3fn, round to nearest even) and one biased E8M0 scale byte per
    32 consecutive elements.  Block rule of the product (csrc/mxfp8.h), restated: the
    smallest power of two 2^e with amax <= 448 * 2^e:  amax = m 2^x, 1 <= m < 2  ->
    e = x - 8 + (m > 1.75), biased E = e + 127 clamped to [0, 253]; elements =
    RNE_e4m3(v * 2^-e).  Returns (q float8 [..., K], E uint8 [..., K/32])."""
    x = x.to(torch.float32)
    K = x.shape[-1]
    assert K % 32 == 0
    xb = x.reshape(*x.shape[:-1], K // 32, 32)
    amax = xb.abs().amax(-1).contiguous()
    bits = amax.view(torch.int32)
    E = (((bits >> 23) & 0xff) - 8 + ((bits & 0x7fffff) > 0x600000).to(torch.int32))
    E = E.clamp(0, 253)
    inv = ((254 - E) << 23).to(torch.int32).view(torch.float32)
    q = (xb * inv.unsqueeze(-1)).to(torch.float8_e4m3fn)
    return q.reshape(x.shape), E.to(torch.uint8)


def mx_dequantize(q: torch.Tensor, E: torch.Tensor) -> torch.Tensor:
    K = q.shape[-1]
    scale = torch.pow(torch.tensor(2.0, dtype=torch.float64), E.to(torch.float64) - 127.0)
    v = q.to(torch.float32).to(torch.float64).reshape(*q.shape[:-1], K // 32, 32)
    return (v * scale.unsqueeze(-1)).reshape(q.shape).to(torch.float32)


def mx_round(x: torch.Tensor) -> torch.Tensor:
    """x -> the fp32 values its MXFP8 image represents (exact: e4m3 x power of two)."""
    return mx_dequantize(*mx_quantize(x))


def x6_planes(x: torch.Tensor):
    """The product's exact three-way bf16 split of an fp32 tensor (csrc/x6.h split3,
    restated): x0 = bf16(x) (round to nearest even), x1 = bf16(x - x0), x2 = x - x0 - x1.
    Both subtractions are exact in fp32 and x2 has at most 8 significant bits, so
    x0 + x1 + x2 == x bit for bit (returned as three fp32 tensors holding bf16 values)."""
    x = x.to(torch.float32)
    x0 = x.to(torch.bfloat16).to(torch.float32)
    r1 = x - x0
    x1 = r1.to(torch.bfloat16).to(torch.float32)
    x2 = (r1 - x1).to(torch.bfloat16).to(torch.float32)
    return x0, x1, x2


def x6_matmul(a: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """a @ w.T as the product's six-product fp32 GEMM forms it (csrc/gemm_x6.hip): of the
    nine plane products the six with i + j <= 2 (a0b0, a0b1, a1b0, a1b1, a0b2, a2b0); every
    plane product is exact in fp32 (8 x 8 significand bits).  Evaluated here in fp64 --
    what is restated is WHICH terms are kept, not the accumulation order."""
    ap = [t.double() for t in x6_planes(a)]
    wp = [t.double() for t in x6_planes(w)]
    out = torch.zeros(a.shape[0], w.shape[0], dtype=torch.float64)
    for i, j in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)):
        out += ap[i] @ wp[j].T
    return out


@contextlib.contextmanager
def bf16_operands(sd=None, attention=True, fp8_ffn=False):
    """Run the oracle with the product's bf16-operand arithmetic (see
    _Bf16OperandFunctional).  `sd`: the state_dict, to exempt `linear_pos`.
    fp8_ffn: the WN_PREC_FP8 mode -- the FFN GEMMs (w_1, w_2 of every encoder layer)
    see MXFP8 operands instead (BASELINE.json configs[4])."""
    global F, _MM_ROUND
    exempt = [v for k, v in (sd or {}).items() if k.endswith('linear_pos.weight')]
    mx = [v for k, v in (sd or {}).items()
          if fp8_ffn and k.startswith('encoder.') and
          k.endswith(('w_1.weight', 'w_2.weight'))]
    saved, F = F, _Bf16OperandFunctional(exempt, mx)
    saved_mm, _MM_ROUND = _MM_ROUND, bool(attention)
    try:
        yield
    finally:
        F = saved
        _MM_ROUND = saved_mm

# --------------------------------------------------------------------------
# result record -- wenet/models/transformer/search.py:30-61


class DecodeResult:

    def __init__(self,
                 tokens,
                 score=0.0,
                 confidence=0.0,
                 tokens_confidence=None,
                 times=None,
                 nbest=None,
                 nbest_scores=None,
                 nbest_times=None,
                 text=''):
        self.tokens = tokens
        self.score = score
        self.confidence = confidence
        self.tokens_confidence = tokens_confidence
        self.times = times
        self.nbest = nbest
        self.nbest_scores = nbest_scores
        self.nbest_times = nbest_times
        self.text = text


# --------------------------------------------------------------------------
# features


def _mel_scale(freq):  # runtime/core/frontend/fbank.h:196-213 (HTK)
    return np.float32(1127.0) * np.log(
        np.float32(1.0) + np.float32(freq) / np.float32(700.0),
        dtype=np.float32)


def mel_banks(num_bins=80, sample_rate=16000, fft_points=512, low_freq=20.0):
    """HTK triangular filters, runtime/core/frontend/fbank.h:91-148.

    Returns a dense (num_bins, fft_points//2) fp32 matrix (the reference stores
    each row sparsely as (first_index, weights)).
    """
    num_fft_bins = fft_points // 2
    fft_bin_width = np.float32(sample_rate) / np.float32(fft_points)
    mel_low = _mel_scale(low_freq)
    mel_high = _mel_scale(sample_rate / 2)
    delta = (mel_high - mel_low) / np.float32(num_bins + 1)
    out = np.zeros((num_bins, num_fft_bins), dtype=np.float32)
    for b in range(num_bins):
        left = np.float32(mel_low + np.float32(b) * delta)
        center = np.float32(mel_low + np.float32(b + 1) * delta)
        right = np.float32(mel_low + np.float32(b + 2) * delta)
        for i in range(num_fft_bins):
            mel = _mel_scale(fft_bin_width * np.float32(i))
            if mel > left and mel < right:
                if mel <= center:
                    w = (mel - left) / (center - left)
                else:
                    w = (right - mel) / (right - center)
                out[b, i] = np.float32(w)
    return out


def povey_window(frame_length=400):
    """runtime/core/frontend/fbank.h:150-156."""
    a = 2.0 * math.pi / (frame_length - 1)
    i = np.arange(frame_length, dtype=np.float64)
    return np.power(0.5 - 0.5 * np.cos(a * i), 0.85).astype(np.float32)


def fbank(waveform: np.ndarray,
          num_mel_bins: int = 80,
          frame_length: int = 400,
          frame_shift: int = 160,
          sample_rate: int = 16000) -> np.ndarray:
    """Kaldi-compatible log-mel fbank of ONE utterance.

    `waveform` is float in [-1, 1] as `torchaudio.load` returns; it is scaled by
    1<<15 exactly like wenet/dataset/processor.py:245 before the Kaldi recipe
    (dither 0, energy_floor 0, povey window, snip_edges, remove DC, pre-emphasis
    0.97, 512-point FFT, HTK mel 20 Hz..Nyquist, log floor FLT_EPSILON), restated
    from runtime/core/frontend/fbank.h:250-327.  -> (T, num_mel_bins) fp32.
    """
    wave = waveform.astype(np.float32) * np.float32(1 << 15)
    n = wave.shape[0]
    if n < frame_length:
        return np.zeros((0, num_mel_bins), dtype=np.float32)
    num_frames = 1 + (n - frame_length) // frame_shift
    fft_points = 1 << int(math.ceil(math.log2(frame_length)))
    idx = (np.arange(num_frames)[:, None] * frame_shift +
           np.arange(frame_length)[None, :])
    data = wave[idx]  # (T, 400)
    # remove DC offset (fbank.h:272-277); float accumulation
    mean = data.sum(axis=1, dtype=np.float32) / np.float32(frame_length)
    data = data - mean[:, None]
    # pre-emphasis (fbank.h:221-226)
    pre = np.empty_like(data)
    pre[:, 1:] = data[:, 1:] - np.float32(0.97) * data[:, :-1]
    pre[:, 0] = data[:, 0] - np.float32(0.97) * data[:, 0]
    pre = pre * povey_window(frame_length)[None, :]
    padded = np.zeros((num_frames, fft_points), dtype=np.float32)
    padded[:, :frame_length] = pre
    spec = np.fft.rfft(padded.astype(np.float64), axis=1)
    power = (spec.real**2 + spec.imag**2)[:, :fft_points // 2].astype(
        np.float32)  # fbank.h:292-294 keeps bins [0, N/2)
    banks = mel_banks(num_mel_bins, sample_rate, fft_points)
    mel = power @ banks.T
    mel = np.maximum(mel, np.finfo(np.float32).eps)  # fbank.h:306
    return np.log(mel).astype(np.float32)


def slaney_mel_filters(sr: int = 16000, n_fft: int = 400, n_mels: int = 80) -> np.ndarray:
    """`librosa.filters.mel(sr, n_fft, n_mels)` with its defaults (fmin 0, fmax
    sr/2, htk=False, norm='slaney') -> (n_mels, 1 + n_fft // 2) fp32.

    librosa is a third-party dependency of the reference (requirements.txt,
    unpinned) that is absent here; this restates its published algorithm
    (librosa/filters.py `mel`, librosa/core/convert.py `mel_frequencies`:
    Slaney's Auditory-Toolbox scale, linear below 1 kHz, log above, area
    normalisation 2 / (f[i+2] - f[i])).  The reference's in-tree C++ frontend
    builds the same triangles (runtime/core/frontend/fbank.h:113-134,179-210),
    on a 512-point FFT grid: tests/test_oracle.py pins THIS function evaluated at
    n_fft = 512 against the reference C++ filter bank (oracle/_ref), i.e. the scale,
    the triangle construction and the area normalisation; only the evaluation grid
    of the 400-point matrix (201 bins at 40 Hz) is not covered by reference code."""
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz,
                        min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep,
                        f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel,
                        min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def log_mel_spectrogram(waveform: np.ndarray, num_mel_bins: int = 80,
                        n_fft: int = 400, hop_length: int = 160,
                        padding: int = 0, pad_or_trim: bool = False,
                        max_duration: int = 30, sample_rate: int = 16000) -> np.ndarray:
    """compute_log_mel_spectrogram, wenet/dataset/processor.py:320-369 for ONE
    utterance (float waveform in [-1, 1]) -> (T, num_mel_bins) fp32.  torch.stft
    is the reference's own STFT call; the mel matrix is slaney_mel_filters."""
    w = torch.as_tensor(np.asarray(waveform, dtype=np.float32))
    if padding > 0:
        w = F.pad(w, (0, padding))
    if pad_or_trim:
        length = max_duration * sample_rate
        w = w[:length] if w.size(0) >= length else F.pad(w, (0, length - w.size(0)))
    window = torch.hann_window(n_fft)
    stft = torch.stft(w, n_fft, hop_length, window=window, return_complex=True)
    magnitudes = stft[..., :-1].abs()**2
    filters = torch.from_numpy(slaney_mel_filters(sample_rate, n_fft, num_mel_bins))
    mel_spec = filters @ magnitudes
    return whisper_log_norm(mel_spec).transpose(0, 1).contiguous().numpy()


def whisper_log_norm(mel_spec: torch.Tensor) -> torch.Tensor:
    """processor.py:364-367: log10 with floor 1e-10, clamp to (max - 8), (x + 4) / 4 --
    the reference C++ frontend's log + WhisperNorm (frontend/fbank.h:236-247,303-312),
    against which tests/test_oracle.py pins it."""
    log_spec = torch.clamp(mel_spec, min=1e-10).log10()
    log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)
    return (log_spec + 4.0) / 4.0


def whisper_frontend_512(waveform: np.ndarray, num_mel_bins: int = 80) -> np.ndarray:
    """The log-mel chain of log_mel_spectrogram assembled the way the reference's C++
    frontend runs it in its Whisper configuration (frontend/feature_pipeline.h:64-72,
    fbank.h:250-327): snip-edges frames of 400 samples, per-frame DC removal (the C++
    default), periodic Hann window, zero-padded 512-point FFT, power of bins 0..255,
    Slaney filters on that grid, then whisper_log_norm.  Built from the SAME pieces
    (torch.hann_window, slaney_mel_filters, whisper_log_norm) so that the reference C++
    output pins them; the 400-point STFT framing of the Python path is not pinnable."""
    x = torch.as_tensor(np.asarray(waveform, dtype=np.float32))
    n = 1 + (x.numel() - 400) // 160
    frames = x.unfold(0, 400, 160)[:n].clone()
    frames = frames - frames.mean(1, keepdim=True)
    frames = frames * torch.hann_window(400)
    spec = torch.fft.rfft(F.pad(frames, (0, 112)), dim=1)
    power = (spec.real ** 2 + spec.imag ** 2)[:, :256]
    filt = torch.from_numpy(slaney_mel_filters(16000, 512, num_mel_bins))[:, :256]
    return whisper_log_norm(filt @ power.t()).t().contiguous().numpy()


def resample(waveform: np.ndarray, orig_freq: int, new_freq: int = 16000) -> np.ndarray:
    """processor.resample (wenet/dataset/processor.py:177-196) =
    torchaudio.transforms.Resample(orig_freq, new_freq)(waveform) with its
    defaults (resampling_method 'sinc_interp_hann', lowpass_filter_width 6,
    rolloff 0.99).

    Pinned (round 3) against a SECOND implementation, not against torchaudio itself:
    torchaudio is a third-party dependency that is neither vendored in /root/reference
    nor installed here (requirements.txt: torchaudio>=2.1.2, unpinned), and the
    reference has no test or golden vector at this call site; tests/golden/resample_*.npz
    (oracle/gen_golden_resample.py) is the sample-by-sample fp64 evaluation of the
    definition the published kernel implements, sharing no table or padding arithmetic
    with this function (agreement: 3e-8 .. 8e-8).  The algorithm is restated from
    torchaudio 2.1's published `_get_sinc_resample_kernel` /
    `_apply_sinc_resample_kernel`: reduce the rates by their gcd; per output
    phase i in [0, new) a windowed-sinc filter of 2*width + orig taps,
    width = ceil(6 * orig / (0.99 * min(orig, new))), evaluated in fp64 and
    stored fp32; zero-pad (width, width + orig); strided correlation (stride
    orig); keep ceil(new * len / orig) samples.  tests/ check it through
    properties (identity, length, DC gain, tone preservation) and against
    scipy.signal.resample_poly on band-limited input."""
    x = np.asarray(waveform, dtype=np.float32)
    if orig_freq == new_freq:
        return x
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    lpw, rolloff = 6, 0.99
    base = min(orig, new) * rolloff
    width = math.ceil(lpw * orig / base)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx
    t = np.clip(t * base, -lpw, lpw)
    window = np.cos(t * math.pi / lpw / 2) ** 2
    t = t * math.pi
    with np.errstate(invalid='ignore', divide='ignore'):
        kernels = np.where(t == 0, 1.0, np.sin(t) / t)
    kernels = (kernels * window * (base / orig)).astype(np.float32)  # (new, K)
    n = x.shape[-1]
    padded = np.concatenate([np.zeros(width, np.float32), x,
                             np.zeros(width + orig, np.float32)])
    K = kernels.shape[1]
    n_win = (padded.shape[0] - K) // orig + 1
    win = np.lib.stride_tricks.sliding_window_view(padded, K)[::orig][:n_win]
    out = (win.astype(np.float32) @ kernels.T).reshape(-1)  # (n_win, new) row-major
    target = -(-new * n // orig)
    return out[:target].astype(np.float32)


def padding(feats: List[np.ndarray]) -> Tuple[torch.Tensor, torch.Tensor, List[int]]:
    """wenet/dataset/processor.py:526-577 (feature part): sort by length
    descending, zero-pad to (B, Tmax, F); returns (padded, lengths, order)."""
    lens = torch.tensor([f.shape[0] for f in feats], dtype=torch.int32)
    order = torch.argsort(lens, descending=True)
    sorted_feats = [torch.as_tensor(feats[i]) for i in order]
    padded = torch.nn.utils.rnn.pad_sequence(sorted_feats,
                                             batch_first=True,
                                             padding_value=0)
    return padded, lens[order], order.tolist()


# --------------------------------------------------------------------------
# masks -- wenet/utils/mask.py


def make_pad_mask(lengths: torch.Tensor, max_len: int = 0) -> torch.Tensor:
    """wenet/utils/mask.py:201-227: True at padded positions."""
    max_len = max_len if max_len > 0 else int(lengths.max().item())
    seq = torch.arange(0, max_len, dtype=torch.int64)
    return seq.unsqueeze(0) >= lengths.to(torch.int64).unsqueeze(-1)


def subsequent_chunk_mask(size: int, chunk_size: int,
                          num_left_chunks: int = -1) -> torch.Tensor:
    """wenet/utils/mask.py:88-123."""
    i = torch.arange(size)
    ending = torch.clamp((i // chunk_size + 1) * chunk_size, max=size)
    if num_left_chunks < 0:
        start = torch.zeros_like(i)
    else:
        start = torch.clamp((i // chunk_size - num_left_chunks) * chunk_size,
                            min=0)
    j = torch.arange(size).unsqueeze(0)
    return (j >= start.unsqueeze(1)) & (j < ending.unsqueeze(1))


def subsequent_mask(size: int) -> torch.Tensor:
    """wenet/utils/mask.py:51-85: lower-triangular (size, size) bool."""
    a = torch.arange(size)
    return a.unsqueeze(0) <= a.unsqueeze(1)


def add_optional_chunk_mask(xs, masks, use_dynamic_chunk, decoding_chunk_size,
                            static_chunk_size, num_decoding_left_chunks):
    """wenet/utils/mask.py:126-198, decode-time branches only
    (decoding_chunk_size != 0 is asserted by ASRModel.decode,
    asr_model.py:310, so the random training branch is unreachable)."""
    if use_dynamic_chunk:
        max_len = xs.size(1)
        if decoding_chunk_size < 0:
            chunk_size, num_left = max_len, -1
        else:
            assert decoding_chunk_size > 0
            chunk_size, num_left = decoding_chunk_size, num_decoding_left_chunks
        cm = subsequent_chunk_mask(xs.size(1), chunk_size, num_left)
        return masks & cm.unsqueeze(0)
    if static_chunk_size > 0:
        cm = subsequent_chunk_mask(xs.size(1), static_chunk_size,
                                   num_decoding_left_chunks)
        return masks & cm.unsqueeze(0)
    return masks


# --------------------------------------------------------------------------
# encoder building blocks


def global_cmvn(x, mean, istd):
    """wenet/models/transformer/cmvn.py:36-47."""
    return (x - mean) * istd


def positional_encoding_table(d_model: int, max_len: int = 5000):
    """wenet/models/transformer/embedding.py:47-56 (the `pe` buffer)."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(
        torch.arange(0, d_model, 2, dtype=torch.float32) *
        -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0)


def conv2d_subsampling4(x, x_mask, sd, pfx, d_model):
    """wenet/models/transformer/subsampling.py:203-228 +
    RelPositionalEncoding.forward embedding.py:134-147."""
    x = x.unsqueeze(1)
    x = F.relu(F.conv2d(x, sd[pfx + 'conv.0.weight'], sd[pfx + 'conv.0.bias'],
                        stride=2))
    x = F.relu(F.conv2d(x, sd[pfx + 'conv.2.weight'], sd[pfx + 'conv.2.bias'],
                        stride=2))
    b, c, t, f = x.size()
    x = F.linear(x.transpose(1, 2).contiguous().view(b, t, c * f),
                 sd[pfx + 'out.0.weight'], sd[pfx + 'out.0.bias'])
    pe = sd.get(pfx + 'pos_enc.pe')
    if pe is None:
        pe = positional_encoding_table(d_model)
    x = x * math.sqrt(d_model)
    pos_emb = pe[:, 0:x.size(1)]
    return x, pos_emb, x_mask[:, :, 2::2][:, :, 2::2]


def layer_norm(x, sd, pfx, eps=1e-5):
    return F.layer_norm(x, (x.size(-1), ), sd[pfx + 'weight'],
                        sd[pfx + 'bias'], eps)


def feed_forward(x, sd, pfx, activation):
    """wenet/models/transformer/positionwise_feed_forward.py:50-58."""
    return F.linear(activation(F.linear(x, sd[pfx + 'w_1.weight'],
                                        sd[pfx + 'w_1.bias'])),
                    sd[pfx + 'w_2.weight'], sd[pfx + 'w_2.bias'])


def _forward_attention(value, scores, mask, sd, pfx, h, d_k):
    """wenet/models/transformer/attention.py:133-178."""
    if mask.size(-1) > 0:
        m = mask.unsqueeze(-3).eq(0)
        m = m[..., :scores.size(-1)]
        scores = scores.masked_fill(m, -float('inf'))
        attn = torch.softmax(scores.float(), dim=-1).masked_fill(m, 0.0)
    else:
        attn = torch.softmax(scores.float(), dim=-1)
    x = _mm(attn, value)
    x = x.transpose(-3, -2).contiguous()
    x = x.view(x.size()[:-2] + (h * d_k, ))
    return F.linear(x, sd[pfx + 'linear_out.weight'],
                    sd[pfx + 'linear_out.bias'])


def _qkv(query, key, value, sd, pfx, h):
    """wenet/models/transformer/attention.py:109-131 (head-first views)."""
    def proj(name, x):
        y = F.linear(x, sd[pfx + name + '.weight'], sd.get(pfx + name + '.bias'))
        y = y.view(y.size()[:-1] + (h, y.size(-1) // h))
        return y.transpose(-3, -2)
    return proj('linear_q', query), proj('linear_k', key), proj('linear_v', value)


def rel_pos_mha(x, mask, pos_emb, sd, pfx, h):
    """RelPositionMultiHeadedAttention.forward,
    wenet/models/transformer/attention.py:364-438 (non-sdpa path, no cache;
    rel_shift is disabled in the reference, :407-409)."""
    d_k = x.size(-1) // h
    q, k, v = _qkv(x, x, x, sd, pfx, h)
    q = q.transpose(1, 2)  # (b, t, h, d_k)
    p = F.linear(pos_emb, sd[pfx + 'linear_pos.weight'])
    p = p.view(pos_emb.size(0), -1, h, d_k).transpose(1, 2)
    q_u = (q + sd[pfx + 'pos_bias_u']).transpose(1, 2)
    q_v = (q + sd[pfx + 'pos_bias_v']).transpose(1, 2)
    matrix_bd = _mm(q_v, p.transpose(-2, -1))
    matrix_ac = _mm(q_u, k.transpose(-2, -1))
    scores = (matrix_ac + matrix_bd) / math.sqrt(d_k)
    return _forward_attention(v, scores, mask, sd, pfx, h, d_k)


def mha(query, key, value, mask, sd, pfx, h):
    """MultiHeadedAttention.forward / MultiHeadedCrossAttention.forward without
    caches, wenet/models/transformer/attention.py:247-304,456-520."""
    d_k = query.size(-1) // h
    q, k, v = _qkv(query, key, value, sd, pfx, h)
    scores = _mm(q, k.transpose(-2, -1)) / math.sqrt(d_k)
    return _forward_attention(v, scores, mask, sd, pfx, h, d_k)


def _conv_norm(x, sd, pfx, activation):
    """The norm + activation in the middle of the conv module
    (convolution.py:139-143) on x (B, C, T): LayerNorm over channels
    (cnn_module_norm 'layer_norm') or eval-mode BatchNorm1d with its running
    statistics ('batch_norm', the default; recognised by its buffers)."""
    c = x.size(1)
    if pfx + 'norm.running_mean' in sd:
        return activation(F.batch_norm(x, sd[pfx + 'norm.running_mean'],
                                       sd[pfx + 'norm.running_var'],
                                       sd[pfx + 'norm.weight'], sd[pfx + 'norm.bias'],
                                       False, 0.0, 1e-5))
    y = F.layer_norm(x.transpose(1, 2), (c, ), sd[pfx + 'norm.weight'],
                     sd[pfx + 'norm.bias'], 1e-5)
    return activation(y).transpose(1, 2)


def conv_module(x, mask_pad, sd, pfx, kernel_size, causal, activation):
    """ConvolutionModule.forward, wenet/models/transformer/convolution.py:98-153
    without a streaming cache."""
    x = x.transpose(1, 2)
    if mask_pad.size(2) > 0:
        x = x.masked_fill(~mask_pad, 0.0)
    lorder = kernel_size - 1 if causal else 0
    if lorder > 0:
        x = F.pad(x, (lorder, 0), 'constant', 0.0)
    x = F.conv1d(x, sd[pfx + 'pointwise_conv1.weight'],
                 sd[pfx + 'pointwise_conv1.bias'])
    x = F.glu(x, dim=1)
    c = x.size(1)
    x = F.conv1d(x, sd[pfx + 'depthwise_conv.weight'],
                 sd[pfx + 'depthwise_conv.bias'],
                 padding=0 if causal else (kernel_size - 1) // 2, groups=c)
    x = _conv_norm(x, sd, pfx, activation)
    x = F.conv1d(x, sd[pfx + 'pointwise_conv2.weight'],
                 sd[pfx + 'pointwise_conv2.bias'])
    if mask_pad.size(2) > 0:
        x = x.masked_fill(~mask_pad, 0.0)
    return x.transpose(1, 2)


def conformer_layer(x, mask, pos_emb, mask_pad, sd, pfx, h, kernel_size,
                    causal):
    """ConformerEncoderLayer.forward, encoder_layer.py:188-265
    (normalize_before=True, macaron style, ff_scale 0.5, swish = SiLU per
    wenet/utils/class_utils.py:42)."""
    act = F.silu
    residual = x
    x = residual + 0.5 * feed_forward(
        layer_norm(x, sd, pfx + 'norm_ff_macaron.'), sd,
        pfx + 'feed_forward_macaron.', act)
    residual = x
    x = residual + rel_pos_mha(layer_norm(x, sd, pfx + 'norm_mha.'), mask,
                               pos_emb, sd, pfx + 'self_attn.', h)
    residual = x
    x = residual + conv_module(layer_norm(x, sd, pfx + 'norm_conv.'), mask_pad,
                               sd, pfx + 'conv_module.', kernel_size, causal,
                               act)
    residual = x
    x = residual + 0.5 * feed_forward(layer_norm(x, sd, pfx + 'norm_ff.'), sd,
                                      pfx + 'feed_forward.', act)
    return layer_norm(x, sd, pfx + 'norm_final.')


def conv1d_subsampling2(x, x_mask, sd, pfx):
    """Conv1dSubsampling2.forward, wenet/models/transformer/subsampling.py:145-171
    + WhisperPositionalEncoding (embedding.py:150-164: xscale = 1, `pe` buffer of
    1500 rows = [sin | cos])."""
    time = x.size(1)
    x = x.transpose(1, 2)
    x = F.gelu(F.conv1d(x, sd[pfx + 'conv.0.weight'], sd[pfx + 'conv.0.bias'],
                        padding=1))
    x = F.gelu(F.conv1d(x, sd[pfx + 'conv.2.weight'], sd[pfx + 'conv.2.bias'],
                        stride=2, padding=1))
    x = x.transpose(1, 2)
    pe = sd[pfx + 'pos_enc.pe']
    pos_emb = pe[:, 0:x.size(1)]
    x = x * 1.0 + pos_emb
    return x, pos_emb, x_mask[:, :, (time + 1) % 2::2]


def transformer_layer(x, mask, sd, pfx, h, activation):
    """TransformerEncoderLayer.forward, encoder_layer.py:94-127
    (normalize_before=True)."""
    x = x + mha(*([layer_norm(x, sd, pfx + 'norm1.')] * 3), mask, sd,
                pfx + 'self_attn.', h)
    return x + feed_forward(layer_norm(x, sd, pfx + 'norm2.'), sd,
                            pfx + 'feed_forward.', activation)


def whisper_positional_table(d_model: int, max_len: int = 1500):
    """The `pe` buffer of WhisperPositionalEncoding, embedding.py:154-163."""
    inc = np.log(10000) / (d_model // 2 - 1)
    inv = torch.exp(-inc * torch.arange(d_model // 2))
    st = torch.arange(max_len)[:, np.newaxis] * inv[np.newaxis, :]
    return torch.cat([torch.sin(st), torch.cos(st)], dim=1).unsqueeze(0)


def encoder_forward(configs: dict,
                    sd: Dict[str, torch.Tensor],
                    xs: torch.Tensor,
                    xs_lens: torch.Tensor,
                    decoding_chunk_size: int = -1,
                    num_decoding_left_chunks: int = -1,
                    return_layers: bool = False):
    """BaseEncoder.forward for a ConformerEncoder,
    wenet/models/transformer/encoder.py:122-181 -> (xs (B,T',d), masks (B,1,T'))."""
    ec = configs['encoder_conf']
    d = ec.get('output_size', 256)
    h = ec.get('attention_heads', 4)
    nblocks = ec.get('num_blocks', 6)
    ksize = ec.get('cnn_module_kernel', 15)
    causal = ec.get('causal', False)
    T = xs.size(1)
    masks = ~make_pad_mask(xs_lens, T).unsqueeze(1)
    if 'encoder.global_cmvn.mean' in sd:
        xs = global_cmvn(xs, sd['encoder.global_cmvn.mean'],
                         sd['encoder.global_cmvn.istd'])
    if configs.get('encoder', 'conformer') == 'transformer':
        # TransformerEncoder as configured by the Whisper recipes
        # (encoder.py:365-437: conv1d2 + abs_pos_whisper + gelu, full context)
        assert ec.get('input_layer') == 'conv1d2' and \
            ec.get('activation_type') == 'gelu'
        xs, pos_emb, masks = conv1d_subsampling2(xs, masks, sd, 'encoder.embed.')
        layers = [xs]
        for i in range(nblocks):
            xs = transformer_layer(xs, masks, sd, f'encoder.encoders.{i}.', h,
                                   F.gelu)
            layers.append(xs)
        xs = layer_norm(xs, sd, 'encoder.after_norm.')
        if return_layers:
            return xs, masks, layers
        return xs, masks
    xs, pos_emb, masks = conv2d_subsampling4(xs, masks, sd, 'encoder.embed.', d)
    mask_pad = masks
    chunk_masks = add_optional_chunk_mask(
        xs, masks, ec.get('use_dynamic_chunk', False), decoding_chunk_size,
        ec.get('static_chunk_size', 0), num_decoding_left_chunks)
    layers = [xs]
    for i in range(nblocks):
        xs = conformer_layer(xs, chunk_masks, pos_emb, mask_pad, sd,
                             f'encoder.encoders.{i}.', h, ksize, causal)
        layers.append(xs)
    xs = layer_norm(xs, sd, 'encoder.after_norm.')
    if return_layers:
        return xs, masks, layers
    return xs, masks


# --------------------------------------------------------------------------
# streaming: one chunk at a time with attention / convolution caches


def rel_pos_mha_cached(x, pos_emb, sd, pfx, h, k_cache, v_cache):
    """RelPositionMultiHeadedAttention.forward with a key/value cache and the
    all-ones mask of forward_chunk (attention.py:180-245,364-438): the new keys
    and values are appended to the cached ones; every key is visible."""
    d_k = x.size(-1) // h
    q, k, v = _qkv(x, x, x, sd, pfx, h)
    if k_cache is not None and k_cache.size(0) > 0:
        k = torch.cat([k_cache, k], dim=2)
        v = torch.cat([v_cache, v], dim=2)
    q = q.transpose(1, 2)
    p = F.linear(pos_emb, sd[pfx + 'linear_pos.weight'])
    p = p.view(pos_emb.size(0), -1, h, d_k).transpose(1, 2)
    q_u = (q + sd[pfx + 'pos_bias_u']).transpose(1, 2)
    q_v = (q + sd[pfx + 'pos_bias_v']).transpose(1, 2)
    scores = (_mm(q_u, k.transpose(-2, -1)) +
              _mm(q_v, p.transpose(-2, -1))) / math.sqrt(d_k)
    out = _forward_attention(v, scores, torch.ones((0, 0, 0), dtype=torch.bool),
                             sd, pfx, h, d_k)
    return out, (k, v)


def conv_module_cached(x, sd, pfx, kernel_size, causal, activation, cache):
    """ConvolutionModule.forward with its left-context cache and no pad mask
    (convolution.py:98-153): for a causal module the cached `lorder` input
    frames (zeros for the first chunk) are put in front BEFORE pointwise_conv1,
    and the last `lorder` frames of that extended input are the new cache."""
    x = x.transpose(1, 2)
    lorder = kernel_size - 1 if causal else 0
    if lorder > 0:
        if cache is None or cache.size(2) == 0:
            x = F.pad(x, (lorder, 0), 'constant', 0.0)
        else:
            x = torch.cat((cache, x), dim=2)
        new_cache = x[:, :, -lorder:]
    else:
        new_cache = torch.zeros((0, 0, 0), dtype=x.dtype)
    x = F.glu(F.conv1d(x, sd[pfx + 'pointwise_conv1.weight'],
                       sd[pfx + 'pointwise_conv1.bias']), dim=1)
    c = x.size(1)
    x = F.conv1d(x, sd[pfx + 'depthwise_conv.weight'],
                 sd[pfx + 'depthwise_conv.bias'],
                 padding=0 if causal else (kernel_size - 1) // 2, groups=c)
    x = _conv_norm(x, sd, pfx, activation)
    x = F.conv1d(x, sd[pfx + 'pointwise_conv2.weight'],
                 sd[pfx + 'pointwise_conv2.bias'])
    return x.transpose(1, 2), new_cache


def forward_chunk(configs, sd, xs, offset: int, required_cache_size: int,
                  att_cache=None, cnn_cache=None):
    """BaseEncoder.forward_chunk for a ConformerEncoder, encoder.py:204-285
    (== ASRModel.forward_encoder_chunk, asr_model.py:385-427).  xs (1, time, F);
    att_cache (L, heads, cache_t1, 2 d_k) or None/empty; cnn_cache
    (L, 1, d, lorder) or None/empty -> (ys (1, chunk, d), new_att_cache,
    new_cnn_cache)."""
    ec = configs['encoder_conf']
    d = ec.get('output_size', 256)
    h = ec.get('attention_heads', 4)
    nblocks = ec.get('num_blocks', 6)
    ksize = ec.get('cnn_module_kernel', 15)
    causal = ec.get('causal', False)
    assert xs.size(0) == 1
    if att_cache is None:
        att_cache = torch.zeros(0, 0, 0, 0)
    if cnn_cache is None:
        cnn_cache = torch.zeros(0, 0, 0, 0)
    masks = torch.ones(1, 1, xs.size(1), dtype=torch.bool)
    xs = global_cmvn(xs, sd['encoder.global_cmvn.mean'], sd['encoder.global_cmvn.istd'])
    xs, _, _ = conv2d_subsampling4(xs, masks, sd, 'encoder.embed.', d)
    cache_t1 = att_cache.size(2)
    chunk = xs.size(1)
    key_size = cache_t1 + chunk
    pe = sd.get('encoder.embed.pos_enc.pe')
    if pe is None:
        pe = positional_encoding_table(d)
    pos_emb = pe[:, offset - cache_t1:offset - cache_t1 + key_size]  # embedding.py:112-132
    if required_cache_size < 0:
        next_start = 0
    elif required_cache_size == 0:
        next_start = key_size
    else:
        next_start = max(key_size - required_cache_size, 0)
    r_att, r_cnn = [], []
    dk = d // h
    for i in range(nblocks):
        pfx = f'encoder.encoders.{i}.'
        if att_cache.size(0) > 0:
            kc, vc = att_cache[i:i + 1, :, :, :dk], att_cache[i:i + 1, :, :, dk:]
        else:
            kc = vc = None
        cc = cnn_cache[i] if cnn_cache.size(0) > 0 else None
        # ConformerEncoderLayer.forward with caches, encoder_layer.py:188-265
        x = xs + 0.5 * feed_forward(layer_norm(xs, sd, pfx + 'norm_ff_macaron.'), sd,
                                    pfx + 'feed_forward_macaron.', F.silu)
        att, (k, v) = rel_pos_mha_cached(layer_norm(x, sd, pfx + 'norm_mha.'), pos_emb,
                                         sd, pfx + 'self_attn.', h, kc, vc)
        x = x + att
        cv, new_cnn = conv_module_cached(layer_norm(x, sd, pfx + 'norm_conv.'), sd,
                                         pfx + 'conv_module.', ksize, causal, F.silu, cc)
        x = x + cv
        x = x + 0.5 * feed_forward(layer_norm(x, sd, pfx + 'norm_ff.'), sd,
                                   pfx + 'feed_forward.', F.silu)
        xs = layer_norm(x, sd, pfx + 'norm_final.')
        r_att.append(torch.cat((k, v), dim=-1)[:, :, next_start:, :])
        r_cnn.append(new_cnn.unsqueeze(0))
    xs = layer_norm(xs, sd, 'encoder.after_norm.')
    return xs, torch.cat(r_att, dim=0), torch.cat(r_cnn, dim=0)


def forward_chunk_by_chunk(configs, sd, xs, decoding_chunk_size: int,
                           num_decoding_left_chunks: int = -1):
    """BaseEncoder.forward_chunk_by_chunk, encoder.py:287-362: overlapping
    feature windows, one forward_chunk each."""
    assert decoding_chunk_size > 0
    subsampling, context = 4, 7  # Conv2dSubsampling4: rate 4, right_context 6 (+1)
    stride = subsampling * decoding_chunk_size
    window = (decoding_chunk_size - 1) * subsampling + context
    n = xs.size(1)
    att_cache = cnn_cache = None
    outs, offset = [], 0
    required = decoding_chunk_size * num_decoding_left_chunks
    for cur in range(0, n - context + 1, stride):
        y, att_cache, cnn_cache = forward_chunk(configs, sd, xs[:, cur:min(cur + window, n)],
                                                offset, required, att_cache, cnn_cache)
        outs.append(y)
        offset += y.size(1)
    ys = torch.cat(outs, 1)
    return ys, torch.ones((1, 1, ys.size(1)), dtype=torch.bool)


# --------------------------------------------------------------------------
# CTC head and searches


def ctc_logprobs(sd, encoder_out, blank_penalty: float = 0.0,
                 blank_id: int = 0):
    """ASRModel.ctc_logprobs asr_model.py:254-265 / CTC.log_softmax ctc.py:73-81."""
    logits = F.linear(encoder_out, sd['ctc.ctc_lo.weight'],
                      sd['ctc.ctc_lo.bias'])
    if blank_penalty > 0.0:
        logits[:, :, blank_id] -= blank_penalty
    return logits.log_softmax(dim=2)


def remove_duplicates_and_blank(hyp: List[int], blank_id: int = 0) -> List[int]:
    """wenet/utils/ctc_utils.py:23-33."""
    new_hyp = []
    cur = 0
    while cur < len(hyp):
        if hyp[cur] != blank_id:
            new_hyp.append(hyp[cur])
        prev = cur
        while cur < len(hyp) and hyp[cur] == hyp[prev]:
            cur += 1
    return new_hyp


def ctc_greedy_search(ctc_probs, ctc_lens, blank_id: int = 0):
    """wenet/models/transformer/search.py:109-124."""
    batch_size, maxlen = ctc_probs.shape[0], ctc_probs.size(1)
    _, topk_index = ctc_probs.topk(1, dim=2)
    topk_index = topk_index.view(batch_size, maxlen)
    mask = make_pad_mask(ctc_lens, maxlen)
    topk_index = topk_index.masked_fill(mask, blank_id)
    return [
        DecodeResult(remove_duplicates_and_blank(hyp.tolist(), blank_id))
        for hyp in topk_index
    ]


def log_add(*args) -> float:
    """wenet/utils/common.py:302-310."""
    if all(a == -float('inf') for a in args):
        return -float('inf')
    a_max = max(args)
    lsp = math.log(sum(math.exp(a - a_max) for a in args))
    return a_max + lsp


class ContextGraph:
    """wenet/utils/context_graph.py:101-265 restated over flat arrays: a trie of
    the biasing phrases (token-id lists) with Aho-Corasick fail arcs.  Node 0 is
    the root.  Per node: `edges[n]` {token: child}, `fail[n]`, `node_score[n]`
    (bonus accumulated from the root), `output_score[n]` (bonus of every phrase
    that ends at n or at a suffix of n), `is_end[n]`.

    Reference quirks that are kept on purpose (they change scores):
      * `is_end` is decided when a node is CREATED (context_graph.py:160-171): a
        phrase that is a proper prefix of an EARLIER phrase never becomes an
        end node;
      * the fail search stops at the first root it reaches (:196-202) and the
        output arc is the nearest `is_end` node on the fail chain (:205-212).
    """

    def __init__(self, context_list: List[List[int]], context_score: float = 6.0):
        self.context_score = context_score
        self.edges = [{}]
        self.fail = [0]
        self.node_score = [0.0]
        self.output_score = [0.0]
        self.token_score = [0.0]
        self.is_end = [False]
        for phrase in context_list:  # context_graph.py:157-172
            n = 0
            for i, tok in enumerate(phrase):
                if tok not in self.edges[n]:
                    last = i == len(phrase) - 1
                    score = self.node_score[n] + context_score
                    self.edges[n][tok] = len(self.edges)
                    self.edges.append({})
                    self.fail.append(0)
                    self.node_score.append(score)
                    self.output_score.append(score if last else 0)
                    self.token_score.append(context_score)
                    self.is_end.append(last)
                n = self.edges[n][tok]
        # breadth-first fill of the fail / output arcs, context_graph.py:175-214
        queue = deque(self.edges[0].values())
        while queue:
            cur = queue.popleft()
            for tok, n in self.edges[cur].items():
                f = self.fail[cur]
                if tok in self.edges[f]:
                    f = self.edges[f][tok]
                else:
                    f = self.fail[f]
                    while tok not in self.edges[f]:
                        f = self.fail[f]
                        if f == 0:
                            break
                    if tok in self.edges[f]:
                        f = self.edges[f][tok]
                self.fail[n] = f
                out = f
                while not self.is_end[out]:
                    out = self.fail[out]
                    if out == 0:
                        out = -1
                        break
                if out >= 0:
                    self.output_score[n] += self.output_score[out]
                queue.append(n)

    @property
    def num_nodes(self):
        return len(self.edges) - 1

    def forward_one_step(self, state: int, token: int) -> Tuple[float, int]:
        """context_graph.py:216-248 -> (bonus, next state)."""
        if token in self.edges[state]:
            n = self.edges[state][token]
            score = self.token_score[n]
        else:
            n = self.fail[state]
            while token not in self.edges[n]:
                n = self.fail[n]
                if n == 0:
                    break
            if token in self.edges[n]:
                n = self.edges[n][token]
            score = self.node_score[n] - self.node_score[state]
        return score + self.output_score[n], n

    def finalize(self, state: int) -> Tuple[float, int]:
        """context_graph.py:250-265."""
        return -self.node_score[state], 0


def tokenize_context(lines: List[str], symbol_table: Dict[str, int]) -> List[List[int]]:
    """context_graph.py:24-58, char units (no BPE model): one phrase per line,
    ' ' -> U+2581, unknown symbols -> <unk> when the table has one."""
    out = []
    for txt in lines:
        labels = []
        for ch in txt.strip():
            ch = '\u2581' if ch == ' ' else ch
            if ch in symbol_table:
                labels.append(symbol_table[ch])
            elif '<unk>' in symbol_table:
                labels.append(symbol_table['<unk>'])
        out.append(labels)
    return out


class PrefixScore:
    """wenet/models/transformer/search.py:64-106."""
    __slots__ = ('s', 'ns', 'v_s', 'v_ns', 'cur_token_prob', 'times_s',
                 'times_ns', 'context_state', 'context_score', 'has_context')

    def __init__(self, s=float('-inf'), ns=float('-inf'), v_s=float('-inf'),
                 v_ns=float('-inf'), context_state=None, context_score=0.0):
        self.s, self.ns, self.v_s, self.v_ns = s, ns, v_s, v_ns
        self.cur_token_prob = float('-inf')
        self.times_s = []
        self.times_ns = []
        self.context_state = context_state
        self.context_score = context_score
        self.has_context = False

    def total_score(self):
        return self.score() + self.context_score

    def copy_context(self, other):
        self.context_score = other.context_score
        self.context_state = other.context_state

    def update_context(self, graph, other, word_id):
        self.copy_context(other)
        score, state = graph.forward_one_step(other.context_state, word_id)
        self.context_score += score
        self.context_state = state

    def score(self):
        return log_add(self.s, self.ns)

    def viterbi_score(self):
        return self.v_s if self.v_s > self.v_ns else self.v_ns

    def times(self):
        return self.times_s if self.v_s > self.v_ns else self.times_ns


def ctc_prefix_beam_search(ctc_probs, ctc_lens, beam_size: int,
                           blank_id: int = 0,
                           context_graph: Optional[ContextGraph] = None,
                           edge_gaps: Optional[list] = None
                           ) -> List[DecodeResult]:
    """wenet/models/transformer/search.py:127-249.

    `edge_gaps` (test instrumentation, not in the reference): a list that receives, per
    utterance, the smallest margin by which any pruning decision of the search was taken --
    min over frames of (score of the last hypothesis kept - score of the first one dropped)
    and of (log-prob of the last token inside topk - first one outside).  A search whose
    margin is below the log-prob tolerance cannot be expected to prune the same way from
    log-probs that differ within that tolerance.

    With a context graph every dict entry takes its (state, bonus) from the
    FIRST contribution that reaches it (`has_context`), the second prune ranks
    by score + bonus, and after the last frame the bonus is REPLACED by
    finalize()'s -node_score(state) (search.py:229-234) without re-sorting.

    Iteration order is the reference's: top-k tokens in torch.topk order
    (descending log-prob), then `cur_hyps` in beam order; `next_hyps` keeps dict
    insertion order and Python's stable sort breaks score ties by it.
    """
    results = []
    for i in range(ctc_probs.shape[0]):
        ctc_prob = ctc_probs[i]
        num_t = int(ctc_lens[i])
        cg = context_graph
        cur_hyps = [(tuple(), PrefixScore(s=0.0, ns=-float('inf'), v_s=0.0,
                                          v_ns=0.0,
                                          context_state=None if cg is None else 0,
                                          context_score=0.0))]
        min_gap = float('inf')
        for t in range(0, num_t):
            logp = ctc_prob[t]
            next_hyps = defaultdict(lambda: PrefixScore())
            _, top_k_index = logp.topk(beam_size)
            if edge_gaps is not None and logp.numel() > beam_size:
                tv = logp.topk(beam_size + 1)[0]
                min_gap = min(min_gap, float(tv[beam_size - 1] - tv[beam_size]))
            for u in top_k_index:
                u = u.item()
                prob = logp[u].item()
                for prefix, ps in cur_hyps:
                    last = prefix[-1] if len(prefix) > 0 else None
                    if u == blank_id:
                        nx = next_hyps[prefix]
                        nx.s = log_add(nx.s, ps.score() + prob)
                        nx.v_s = ps.viterbi_score() + prob
                        nx.times_s = ps.times().copy()
                        if cg is not None and not nx.has_context:
                            nx.copy_context(ps)
                            nx.has_context = True
                    elif u == last:
                        n1 = next_hyps[prefix]
                        n1.ns = log_add(n1.ns, ps.ns + prob)
                        if n1.v_ns < ps.v_ns + prob:
                            n1.v_ns = ps.v_ns + prob
                            if n1.cur_token_prob < prob:
                                n1.cur_token_prob = prob
                                n1.times_ns = ps.times_ns.copy()
                                n1.times_ns[-1] = t
                        if cg is not None and not n1.has_context:
                            n1.copy_context(ps)
                            n1.has_context = True
                        n2 = next_hyps[prefix + (u, )]
                        n2.ns = log_add(n2.ns, ps.s + prob)
                        if n2.v_ns < ps.v_s + prob:
                            n2.v_ns = ps.v_s + prob
                            n2.cur_token_prob = prob
                            n2.times_ns = ps.times_s.copy()
                            n2.times_ns.append(t)
                        if cg is not None and not n2.has_context:
                            n2.update_context(cg, ps, u)
                            n2.has_context = True
                    else:
                        nx = next_hyps[prefix + (u, )]
                        nx.ns = log_add(nx.ns, ps.score() + prob)
                        if nx.v_ns < ps.viterbi_score() + prob:
                            nx.v_ns = ps.viterbi_score() + prob
                            nx.cur_token_prob = prob
                            nx.times_ns = ps.times().copy()
                            nx.times_ns.append(t)
                        if cg is not None and not nx.has_context:
                            nx.update_context(cg, ps, u)
                            nx.has_context = True
            next_hyps = sorted(next_hyps.items(), key=lambda x: x[1].total_score(),
                               reverse=True)
            if edge_gaps is not None and len(next_hyps) > beam_size:
                min_gap = min(min_gap, next_hyps[beam_size - 1][1].total_score() -
                              next_hyps[beam_size][1].total_score())
            cur_hyps = next_hyps[:beam_size]
        if edge_gaps is not None:
            edge_gaps.append(min_gap)
        if cg is not None:
            for _, ps in cur_hyps:
                ps.context_score, ps.context_state = cg.finalize(ps.context_state)
        nbest = [y[0] for y in cur_hyps]
        nbest_scores = [y[1].total_score() for y in cur_hyps]
        nbest_times = [y[1].times() for y in cur_hyps]
        results.append(
            DecodeResult(tokens=nbest[0], score=nbest_scores[0],
                         times=nbest_times[0], nbest=nbest,
                         nbest_scores=nbest_scores, nbest_times=nbest_times))
    return results


# --------------------------------------------------------------------------
# attention decoder + rescoring


def _decoder_forward(configs, sd, pfx, nblocks, memory, memory_mask, ys_in_pad,
                     ys_in_lens):
    """TransformerDecoder.forward decoder.py:146-201 + DecoderLayer.forward
    decoder_layer.py:68-153 (normalize_before, relu FFN, abs-pos embedding
    embedding.py:58-76: x*sqrt(d)+pe)."""
    dc = configs['decoder_conf']
    h = dc.get('attention_heads', 4)
    d = memory.size(-1)
    maxlen = ys_in_pad.size(1)
    tgt_mask = ~make_pad_mask(ys_in_lens, maxlen).unsqueeze(1)
    tgt_mask = tgt_mask & subsequent_mask(maxlen).unsqueeze(0)
    pe = sd.get(pfx + 'embed.1.pe')
    if pe is None:
        pe = positional_encoding_table(d)
    x = F.embedding(ys_in_pad, sd[pfx + 'embed.0.weight']) * math.sqrt(d) + \
        pe[:, :maxlen]
    for j in range(nblocks):
        lp = f'{pfx}decoders.{j}.'
        residual = x
        xn = layer_norm(x, sd, lp + 'norm1.')
        x = residual + mha(xn, xn, xn, tgt_mask, sd, lp + 'self_attn.', h)
        residual = x
        xn = layer_norm(x, sd, lp + 'norm2.')
        x = residual + mha(xn, memory, memory, memory_mask, sd,
                           lp + 'src_attn.', h)
        residual = x
        x = residual + feed_forward(layer_norm(x, sd, lp + 'norm3.'), sd,
                                    lp + 'feed_forward.', F.relu)
    x = layer_norm(x, sd, pfx + 'after_norm.')
    return F.linear(x, sd[pfx + 'output_layer.weight'],
                    sd[pfx + 'output_layer.bias'])


def is_bidirectional(configs) -> bool:
    return configs.get('decoder', 'bitransformer') == 'bitransformer'


def reverse_hyps(hyps, hyps_lens, eos: int):
    """Input of the right-to-left decoder, asr_model.py:485-536: hyps (N, L) start
    with sos and are eos-padded, hyps_lens count the sos; every hypothesis is
    reversed behind its sos and re-padded with eos (worked example in the
    reference's comments, pinned in tests/test_oracle.py)."""
    r_hyps_lens = hyps_lens - 1
    r_hyps = hyps[:, 1:]
    max_len = torch.max(r_hyps_lens)
    index_range = torch.arange(0, max_len, 1)
    seq_len_expand = r_hyps_lens.unsqueeze(1)
    seq_mask = seq_len_expand > index_range
    index = (seq_len_expand - 1) - index_range
    index = index * seq_mask
    r_hyps = torch.gather(r_hyps, 1, index)
    r_hyps = torch.where(seq_mask, r_hyps, eos)
    return torch.cat([hyps[:, 0:1], r_hyps], dim=1)


def forward_attention_decoder(configs, sd, hyps, hyps_lens, encoder_out,
                              reverse_weight: float = 0.0, sos: int = 2,
                              eos: int = 2):
    """ASRModel.forward_attention_decoder asr_model.py:453-547."""
    assert encoder_out.size(0) == 1
    num_hyps = hyps.size(0)
    encoder_out = encoder_out.repeat(num_hyps, 1, 1)
    encoder_mask = torch.ones(num_hyps, 1, encoder_out.size(1),
                              dtype=torch.bool)
    r_hyps = reverse_hyps(hyps, hyps_lens, eos)
    dc = configs['decoder_conf']
    if is_bidirectional(configs):
        lp, nl = 'decoder.left_decoder.', dc.get('num_blocks', 6)
    else:
        lp, nl = 'decoder.', dc.get('num_blocks', 6)
    decoder_out = _decoder_forward(configs, sd, lp, nl, encoder_out,
                                   encoder_mask, hyps, hyps_lens)
    decoder_out = F.log_softmax(decoder_out, dim=-1)
    r_decoder_out = torch.tensor(0.0)
    if is_bidirectional(configs) and reverse_weight > 0.0:
        r_decoder_out = _decoder_forward(configs, sd, 'decoder.right_decoder.',
                                         dc.get('r_num_blocks', 0), encoder_out,
                                         encoder_mask, r_hyps, hyps_lens)
    r_decoder_out = F.log_softmax(r_decoder_out, dim=-1)
    return decoder_out, r_decoder_out


def attention_rescoring(configs, sd, ctc_prefix_results, encoder_outs,
                        encoder_lens, ctc_weight: float = 0.0,
                        reverse_weight: float = 0.0, sos: int = 2,
                        eos: int = 2, ignore_id: int = -1):
    """wenet/models/transformer/search.py:374-458 (non-whisper branch)."""
    results = []
    for b in range(encoder_outs.shape[0]):
        encoder_out = encoder_outs[b, :int(encoder_lens[b]), :].unsqueeze(0)
        hyps = ctc_prefix_results[b].nbest
        ctc_scores = ctc_prefix_results[b].nbest_scores
        hyps_lens = torch.tensor([len(h) for h in hyps], dtype=torch.long)
        maxl = int(hyps_lens.max()) if len(hyps) else 0
        # pad_sequence(..., ignore_id) then add_sos_eos (common.py:113-155):
        # ys_in = [sos] + hyp, padded with eos.
        ys_in = torch.full((len(hyps), maxl + 1), eos, dtype=torch.long)
        ys_in[:, 0] = sos
        for i, h in enumerate(hyps):
            if len(h):
                ys_in[i, 1:1 + len(h)] = torch.tensor(h, dtype=torch.long)
        hyps_lens = hyps_lens + 1
        decoder_out, r_decoder_out = forward_attention_decoder(
            configs, sd, ys_in, hyps_lens, encoder_out, reverse_weight, sos,
            eos)
        best_score, best_index = -float('inf'), 0
        confidences, tokens_confidences, all_scores = [], [], []
        for i, hyp in enumerate(hyps):
            score = 0.0
            tc = []
            for j, w in enumerate(hyp):
                s = decoder_out[i][j][w]
                score += s
                tc.append(math.exp(s))
            score += decoder_out[i][len(hyp)][eos]
            if reverse_weight > 0 and r_decoder_out.dim() > 0:
                r_score = 0.0
                for j, w in enumerate(hyp):
                    s = r_decoder_out[i][len(hyp) - j - 1][w]
                    r_score += s
                    tc[j] = (tc[j] + math.exp(s)) / 2
                r_score += r_decoder_out[i][len(hyp)][eos]
                score = score * (1 - reverse_weight) + r_score * reverse_weight
            confidences.append(math.exp(score / (len(hyp) + 1)))
            score += ctc_scores[i] * ctc_weight
            all_scores.append(float(score))
            if score > best_score:
                best_score = score.item()
                best_index = i
            tokens_confidences.append(tc)
        r = DecodeResult(hyps[best_index], best_score,
                         confidence=confidences[best_index],
                         times=ctc_prefix_results[b].nbest_times[best_index],
                         tokens_confidence=tokens_confidences[best_index])
        r.all_scores = all_scores  # oracle-only extra: every hypothesis' score
        results.append(r)
    return results


def mask_finished_scores(score, flag):
    """wenet/utils/mask.py:258-285."""
    beam_size = score.size(-1)
    zero_mask = torch.zeros_like(flag, dtype=torch.bool)
    if beam_size > 1:
        unfinished = torch.cat((zero_mask, flag.repeat([1, beam_size - 1])), dim=1)
        finished = torch.cat((flag, zero_mask.repeat([1, beam_size - 1])), dim=1)
    else:
        unfinished, finished = zero_mask, flag
    score.masked_fill_(unfinished, -float('inf'))
    score.masked_fill_(finished, 0)
    return score


def attention_beam_search(configs, sd, encoder_out, encoder_mask, beam_size: int = 10,
                          length_penalty: float = 0.0, sos: int = 2, eos: int = 2):
    """attention_beam_search, wenet/models/transformer/search.py:252-371 (the
    non-Whisper branch).  The reference steps the decoder with self/cross
    attention caches (decoder.py:226-281); a cache only avoids recomputation, so
    this restatement runs the full decoder on the growing prefixes and takes
    the last position -- the same numbers."""
    batch_size, maxlen = encoder_out.shape[0], encoder_out.size(1)
    running = batch_size * beam_size
    dc = configs['decoder_conf']
    if is_bidirectional(configs):
        lp, nl = 'decoder.left_decoder.', dc.get('num_blocks', 6)
    else:
        lp, nl = 'decoder.', dc.get('num_blocks', 6)
    memory = encoder_out.unsqueeze(1).repeat(1, beam_size, 1, 1).view(
        running, maxlen, encoder_out.size(2))
    memory_mask = encoder_mask.unsqueeze(1).repeat(1, beam_size, 1, 1).view(
        running, 1, maxlen)
    hyps = torch.ones([running, 1], dtype=torch.long).fill_(sos)
    scores = torch.tensor([0.0] + [-float('inf')] * (beam_size - 1),
                          dtype=torch.float).repeat([batch_size]).unsqueeze(1)
    end_flag = torch.zeros_like(scores, dtype=torch.bool)
    for i in range(1, maxlen + 1):
        if end_flag.sum() == running:
            break
        lens = torch.full((running, ), i, dtype=torch.long)
        out = _decoder_forward(configs, sd, lp, nl, memory, memory_mask, hyps, lens)
        logp = torch.log_softmax(out[:, -1], dim=-1)
        top_k_logp, top_k_index = logp.topk(beam_size)
        top_k_logp = mask_finished_scores(top_k_logp, end_flag)
        top_k_index = top_k_index.masked_fill_(end_flag.repeat([1, beam_size]), eos)
        scores = scores + top_k_logp
        scores = scores.view(batch_size, beam_size * beam_size)
        scores, offset_k_index = scores.topk(k=beam_size)
        scores = scores.view(-1, 1)
        base_k_index = torch.arange(batch_size).view(-1, 1).repeat(
            [1, beam_size]) * beam_size * beam_size
        best_k_index = base_k_index.view(-1) + offset_k_index.view(-1)
        best_k_pred = torch.index_select(top_k_index.view(-1), dim=-1,
                                         index=best_k_index)
        best_hyps_index = best_k_index // beam_size
        last_best_k_hyps = torch.index_select(hyps, dim=0, index=best_hyps_index)
        hyps = torch.cat((last_best_k_hyps, best_k_pred.view(-1, 1)), dim=1)
        end_flag = torch.eq(hyps[:, -1], eos).view(-1, 1)
    scores = scores.view(batch_size, beam_size)
    lengths = hyps.ne(eos).sum(dim=1).view(batch_size, beam_size).float()
    scores = scores / lengths.pow(length_penalty)
    best_scores, best_index = scores.max(dim=-1)
    best_hyps_index = best_index + torch.arange(batch_size, dtype=torch.long) * beam_size
    best_hyps = torch.index_select(hyps, dim=0, index=best_hyps_index)[:, 1:]
    results = []
    for i in range(batch_size):
        hyp = best_hyps[i]
        results.append(DecodeResult(hyp[hyp != eos].tolist()))
    return results


def special_symbols(configs):
    """sos/eos as ASRModel.__init__ resolves them (asr_model.py:52-60): from
    tokenizer_conf.special_tokens if present, else vocab_size-1."""
    vocab = configs['output_dim']
    st = (configs.get('tokenizer_conf') or {}).get('special_tokens') or {}
    return st.get('<sos>', vocab - 1), st.get('<eos>', vocab - 1)


def filter_blank_embedding(ctc_probs, encoder_out, valid_lens=None):
    """ASRModel.filter_blank_embedding, wenet/models/transformer/asr_model.py:153-180: the
    rows of encoder_out (B, T, d) whose CTC arg-max is not token 0, per utterance in order,
    zero-padded to the longest selection; returns (selected (B, T_sel, d), mask (B, 1, T_sel)).
    The reference looks at ALL T frames of every utterance, padded ones included.
    `valid_lens` (not in the reference): only frames t < valid_lens[b] count -- what the
    accelerated path does, which never computes padded frames (identical for one utterance or
    a batch of equal lengths)."""
    B, T = encoder_out.size(0), encoder_out.size(1)
    top1 = torch.argmax(ctc_probs, dim=2)
    sel = []
    for j in range(B):
        n = T if valid_lens is None else int(valid_lens[j])
        idx = torch.tensor([i for i in range(n) if top1[j][i] != 0], dtype=torch.long)
        sel.append(torch.index_select(encoder_out[j], 0, idx))
    out = torch.nn.utils.rnn.pad_sequence(sel, batch_first=True, padding_value=0)
    lens = torch.tensor([x.size(0) for x in sel])
    Ts = out.size(1)
    mask = (torch.arange(Ts).unsqueeze(0) < lens.unsqueeze(1)).unsqueeze(1)
    return out, mask


def decode(configs, sd, methods, speech, speech_lengths, beam_size: int = 1,
           decoding_chunk_size: int = -1, num_decoding_left_chunks: int = -1,
           ctc_weight: float = 0.0, reverse_weight: float = 0.0,
           blank_id: int = 0, blank_penalty: float = 0.0,
           length_penalty: float = 0.0, context_graph=None, nonblank_valid_only=False):
    """ASRModel.decode asr_model.py:267-343 (methods: attention,
    ctc_greedy_search, ctc_prefix_beam_search, attention_rescoring)."""
    assert speech.shape[0] == speech_lengths.shape[0]
    assert decoding_chunk_size != 0
    with torch.no_grad():
        encoder_out, encoder_mask = encoder_forward(configs, sd, speech,
                                                    speech_lengths,
                                                    decoding_chunk_size,
                                                    num_decoding_left_chunks)
        encoder_lens = encoder_mask.squeeze(1).sum(1)
        ctc_probs = ctc_logprobs(sd, encoder_out, blank_penalty, blank_id)
        sos, eos = special_symbols(configs)
        results = {}
        if 'attention' in methods:
            results['attention'] = attention_beam_search(
                configs, sd, encoder_out, encoder_mask, beam_size, length_penalty,
                sos, eos)
        if 'ctc_greedy_search' in methods:
            results['ctc_greedy_search'] = ctc_greedy_search(
                ctc_probs, encoder_lens, blank_id)
        if 'ctc_prefix_beam_search' in methods:
            results['ctc_prefix_beam_search'] = ctc_prefix_beam_search(
                ctc_probs, encoder_lens, beam_size, blank_id, context_graph)
        if 'attention_rescoring' in methods:
            pre = results.get('ctc_prefix_beam_search')
            if pre is None:
                pre = ctc_prefix_beam_search(ctc_probs, encoder_lens, beam_size,
                                             blank_id, context_graph)
            if (configs.get('model_conf') or {}).get('apply_non_blank_embedding', False):
                # asr_model.py:337-342: filtered memory, UNFILTERED lengths (search.py:396
                # then slices the zero-padded tensor, zero rows included)
                encoder_out, _ = filter_blank_embedding(
                    ctc_probs, encoder_out,
                    encoder_lens if nonblank_valid_only else None)
            results['attention_rescoring'] = attention_rescoring(
                configs, sd, pre, encoder_out, encoder_lens, ctc_weight,
                reverse_weight, sos, eos)
    return results
