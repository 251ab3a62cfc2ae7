"""ctypes binding of oracle/_ref/libref_fbank.so -- the REFERENCE's own C++
fbank (runtime/core/frontend/fbank.h) and wav reader (frontend/wav.h) built by
oracle/Makefile.  Test infrastructure only."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, '_ref', 'libref_fbank.so')


def available() -> bool:
    return os.path.exists(LIB_PATH)


_lib = None


def ref_fbank(waveform: np.ndarray, num_bins: int = 80, sample_rate: int = 16000,
              frame_length: int = 400, frame_shift: int = 160) -> np.ndarray:
    """`waveform` float in [-1, 1]; scaled by 1 << 15 like
    wenet/dataset/processor.py:245 before the reference recipe."""
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.ref_fbank.restype = ctypes.c_int
        _lib.ref_fbank.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_void_p, ctypes.c_int]
    w = np.ascontiguousarray(waveform.astype(np.float32) * np.float32(1 << 15))
    n = w.shape[0]
    max_frames = max(1, 1 + max(0, n - frame_length) // frame_shift)
    out = np.zeros((max_frames, num_bins), dtype=np.float32)
    got = _lib.ref_fbank(w.ctypes.data, n, num_bins, sample_rate, frame_length,
                         frame_shift, out.ctypes.data, max_frames)
    assert got >= 0, 'ref_fbank: output buffer too small'
    return out[:got]


def has_whisper_frontend() -> bool:
    if not available():
        return False
    try:
        return hasattr(ctypes.CDLL(LIB_PATH), 'ref_whisper_fbank')
    except OSError:
        return False


def has_wav_reader() -> bool:
    """The built library exports ref_wav_read (a library built before that entry
    point existed does not: rebuild with `make -C oracle`)."""
    if not available():
        return False
    try:
        return hasattr(ctypes.CDLL(LIB_PATH), 'ref_wav_read')
    except OSError:
        return False


def ref_wav_read(path: str):
    """The reference's WavReader: (interleaved raw integer sample values as
    float32, channels, sample rate, bits per sample)."""
    lib = ctypes.CDLL(LIB_PATH)
    lib.ref_wav_read.restype = ctypes.c_int
    lib.ref_wav_read.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int,
                                 ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                 ctypes.POINTER(ctypes.c_int)]
    cap = max(1, os.path.getsize(path))
    out = np.zeros((cap, ), dtype=np.float32)
    ch, sr, bits = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    n = lib.ref_wav_read(path.encode(), out.ctypes.data, cap, ctypes.byref(ch),
                         ctypes.byref(sr), ctypes.byref(bits))
    assert n >= 0, f'ref_wav_read failed ({n})'
    return out[:n], ch.value, sr.value, bits.value


def ref_whisper_fbank(waveform: np.ndarray, num_bins: int = 80) -> np.ndarray:
    """The reference C++ frontend in its Whisper configuration (feature_pipeline.h:64-72)
    on a float waveform in [-1, 1] -> (T, num_bins)."""
    lib = ctypes.CDLL(LIB_PATH)
    x = np.ascontiguousarray(np.asarray(waveform, dtype=np.float32) * 32768.0)
    max_frames = max(1, 1 + (len(x) - 400) // 160) if len(x) >= 400 else 1
    out = np.zeros((max_frames, num_bins), dtype=np.float32)
    lib.ref_whisper_fbank.restype = ctypes.c_int
    lib.ref_whisper_fbank.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_int]
    n = lib.ref_whisper_fbank(x.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), len(x),
                              num_bins, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                              max_frames)
    assert n >= 0
    return out[:n]


def ref_slaney_filters(num_bins: int = 80):
    """(weights (num_bins, 256) on the 512-point FFT grid, periodic Hanning window (400,))
    of the reference C++ frontend's Whisper configuration."""
    lib = ctypes.CDLL(LIB_PATH)
    w = np.zeros((num_bins, 256), dtype=np.float32)
    win = np.zeros((400, ), dtype=np.float32)
    lib.ref_slaney_filters.restype = None
    lib.ref_slaney_filters.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.ref_slaney_filters(num_bins, w.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                           win.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return w, win
