"""GPU operator tests: each HIP kernel family against a plain torch fp32
reference of the same op (run on the box's CPU)."""
import ctypes
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _gemm(A, W, bias=None, resid=None, alpha=1.0, act=0):
    from wenet_amd import _lib
    L = _lib.lib()
    M, K = A.shape
    N = W.shape[0]
    C = torch.empty((M, N), dtype=torch.float32, device='cuda')
    _lib.check(L.wn_op_gemm(_ptr(A), _ptr(W), _ptr(bias), _ptr(resid), _ptr(C),
                            M, N, K, alpha, act,
                            torch.cuda.current_stream().cuda_stream), 'gemm')
    torch.cuda.synchronize()
    return C


@pytest.mark.parametrize('M,N,K', [
    (128, 128, 32), (1, 1, 32), (77, 67, 64), (300, 256, 256),
    (7936, 2048, 256), (7936, 256, 2048), (513, 4233, 256), (2000, 768, 256),
    (129, 130, 2432)])
def test_gemm_plain_asymmetric(M, N, K):
    g = torch.Generator().manual_seed(M * 131 + N * 7 + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g)  # asymmetric operands (transpose check)
    ref = A.double() @ W.double().T
    got = _gemm(A.cuda(), W.cuda()).cpu().double()
    err = (got - ref).abs().max().item()
    scale = (A.abs().double() @ W.abs().double().T).max().item()
    assert err <= 2e-6 * scale, (err, scale)


def test_gemm_identity_detects_transposed_store():
    K = 64
    A = torch.eye(K)
    W = torch.arange(96 * K, dtype=torch.float32).reshape(96, K) / 100.0
    got = _gemm(A.cuda(), W.cuda()).cpu()
    torch.testing.assert_close(got, W.T.contiguous(), rtol=0, atol=1e-6)


@pytest.mark.parametrize('act', [0, 1, 2])
@pytest.mark.parametrize('use_resid', [False, True])
def test_gemm_epilogues(act, use_resid):
    g = torch.Generator().manual_seed(act * 2 + int(use_resid))
    M, N, K = 333, 200, 96
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.2
    bias = torch.randn(N, generator=g)
    resid = torch.randn(M, N, generator=g) if use_resid else None
    y = A @ W.T + bias
    if act == 1:
        y = torch.nn.functional.silu(y)
    elif act == 2:
        y = torch.relu(y)
    y = 0.5 * y
    if use_resid:
        y = y + resid
    got = _gemm(A.cuda(), W.cuda(), bias.cuda(),
                resid.cuda() if use_resid else None, 0.5, act).cpu()
    torch.testing.assert_close(got, y, rtol=1e-5, atol=2e-5)


def test_gemm_in_place_residual():
    g = torch.Generator().manual_seed(5)
    M, N, K = 257, 256, 512
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.1
    x = torch.randn(M, N, generator=g)
    from wenet_amd import _lib
    xc = x.cuda()
    Ac, Wc = A.cuda(), W.cuda()
    _lib.check(_lib.lib().wn_op_gemm(Ac.data_ptr(), Wc.data_ptr(), None,
                                     xc.data_ptr(), xc.data_ptr(), M, N, K,
                                     1.0, 0, None), 'gemm')
    torch.cuda.synchronize()
    torch.testing.assert_close(xc.cpu(), x + A @ W.T, rtol=1e-5, atol=5e-5)


def test_gemm_rejects_bad_k():
    from wenet_amd import _lib
    A = torch.zeros(4, 30, device='cuda')
    W = torch.zeros(4, 30, device='cuda')
    C = torch.zeros(4, 4, device='cuda')
    st = _lib.lib().wn_op_gemm(A.data_ptr(), W.data_ptr(), None, None,
                               C.data_ptr(), 4, 4, 30, 1.0, 0, None)
    assert st != 0 and b'multiple of 32' in _lib.lib().wn_last_error()


@pytest.mark.parametrize('D,M', [(64, 1001), (128, 1001), (256, 1001), (512, 1001),
                                 (256, 7933), (512, 4097), (1280, 5000)])
def test_layernorm(D, M):
    """One row per wave, odd M = a partly filled last block."""
    from wenet_amd import _lib
    g = torch.Generator().manual_seed(D)
    x = torch.randn(M, D, generator=g) * 3 + 1
    w, b = torch.randn(D, generator=g), torch.randn(D, generator=g)
    ref = torch.nn.functional.layer_norm(x, (D, ), w, b, 1e-5)
    xc, wc, bc = x.cuda(), w.cuda(), b.cuda()
    y = torch.empty_like(xc)
    _lib.check(_lib.lib().wn_op_layernorm(xc.data_ptr(), wc.data_ptr(),
                                          bc.data_ptr(), y.data_ptr(), M, D,
                                          1e-5, None), 'ln')
    torch.cuda.synchronize()
    torch.testing.assert_close(y.cpu(), ref, rtol=1e-5, atol=2e-5)


def test_log_add_matches_reference_formula():
    """wn_op_log_add (the prefix beam search's fp64 log_add) against
    wenet/utils/common.py:302-310 evaluated with Python floats."""
    from wenet_amd import _lib
    L = _lib.lib()
    rng = np.random.Generator(np.random.PCG64(7))
    n = 20000
    a = -rng.random(n) * 60.0
    # gaps from 1e-12 to 60, both orders, plus the special cases
    gap = rng.random(n) * 10.0 ** rng.integers(-12, 2, size=n)
    b = a - gap
    swap = rng.random(n) < 0.5
    a, b = np.where(swap, b, a), np.where(swap, a, b)
    a[:6] = [-np.inf, -np.inf, -3.5, 0.0, -1.0, -40.0]
    b[:6] = [-np.inf, -2.25, -np.inf, 0.0, -38.5, -2.0]
    ta = torch.from_numpy(a).cuda()
    tb = torch.from_numpy(b).cuda()
    to = torch.empty_like(ta)
    _lib.check(L.wn_op_log_add(_ptr(ta), _ptr(tb), _ptr(to), n,
                               torch.cuda.current_stream().cuda_stream), 'log_add')
    got = to.cpu().numpy()

    def ref(x, y):
        if x == -math.inf and y == -math.inf:
            return -math.inf
        m = max(x, y)
        return m + math.log(math.exp(x - m) + math.exp(y - m))
    want = np.array([ref(float(x), float(y)) for x, y in zip(a, b)])
    assert got[0] == -np.inf and got[1] == -2.25 and got[2] == -3.5
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin)
    err = np.abs(got[fin] - want[fin])
    # the result is a_max + log(...): its rounding error scales with the larger
    # of |a_max| and |result| (a_max ~ -log(...) cancels to a small result)
    scale = np.maximum(np.abs(want[fin]), np.abs(np.maximum(a, b)[fin]))
    ulp = np.spacing(np.maximum(scale, 1e-300))
    assert (err <= 4 * ulp).all(), (err / ulp).max()


@pytest.mark.parametrize('orig,new,n', [(8000, 16000, 12345), (44100, 16000, 50001),
                                        (48000, 16000, 30000), (22050, 16000, 7),
                                        (16000, 8000, 16001), (11025, 16000, 9999),
                                        (16000, 16000, 500)])
def test_resample_vs_oracle(orig, new, n):
    """wn_resample (processor.resample; torchaudio Resample defaults) against the
    oracle's restatement: same taps (fp64 -> fp32), fp32 accumulation."""
    import ctypes
    from oracle import wenet_oracle as O
    from gpu_util import cached_model
    _, _, model = cached_model('tiny_sym', 0)
    rng = np.random.RandomState(orig % 97 + n)
    x = (rng.rand(n).astype(np.float32) * 2 - 1) * 0.7
    got = model.resample(x, orig, new)
    want = O.resample(x, orig, new)
    assert got.shape == want.shape and got.dtype == np.float32
    assert np.abs(got - want).max() < 2e-5, np.abs(got - want).max()
    L = model._L
    assert L.wn_resample_length(n, orig, new) == want.shape[0]


@pytest.mark.parametrize('orig,new', [(44100, 16000), (48000, 16000), (8000, 16000),
                                      (22050, 16000), (16000, 8000), (11025, 16000),
                                      (32000, 16000)])
def test_resample_vs_the_independent_fp64_definition(orig, new):
    """wn_resample against tests/golden/resample_*.npz: the sample-by-sample fp64 evaluation of
    the published torchaudio resampling definition (oracle/gen_golden_resample.py) -- a second
    implementation that shares no code or intermediate table with the oracle's polyphase
    restatement.  Bound: fp32 taps and fp32 accumulation of ~30 terms of magnitude <= 0.3."""
    import os
    from gpu_util import cached_model
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', f'resample_{orig}_{new}.npz'))
    _, _, model = cached_model('tiny_sym', 0)
    got = model.resample(z['x'], orig, new)
    assert got.shape == z['y'].shape
    err = np.abs(got.astype(np.float64) - z['y']).max()
    assert err < 5e-6, err
