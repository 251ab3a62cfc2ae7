"""The six-product fp32 GEMM (csrc/gemm_x6.hip: fp32 operands as three exact bf16 planes,
six bf16 MFMA products, fp32 accumulate) through the C ABI against fp64, next to the
v_mfma_f32 kernel on the same inputs: the claim under test is that it is an fp32 GEMM --
its error against fp64 is not larger than the fp32 matrix-core kernel's."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[1, 0], ids=['a_fp32', 'a_planes'], autouse=True)
def a_operand_form(request):
    """Both forms of the A operand: a plane image made by x6_split / a GEMM epilogue (the
    default) and plain fp32 rows split in registers."""
    from wenet_amd import _lib
    L = _lib.lib()
    _lib.check(L.wn_tune_set(b'x6_af32', request.param), 'tune')
    yield
    L.wn_tune_set(b'x6_af32', 0)


def _x6(A, W, b, r, act, alpha, bm=0):
    from wenet_amd import _lib
    L = _lib.lib()
    M, K = A.shape
    N = W.shape[0]
    C = torch.empty((M, N), device='cuda')
    _lib.check(L.wn_op_gemm_x6(A.data_ptr(), W.data_ptr(), b.data_ptr() if b is not None else None,
                               r.data_ptr() if r is not None else None, C.data_ptr(), M, N, K,
                               alpha, act, bm, 1, torch.cuda.current_stream().cuda_stream),
               'gemm_x6')
    torch.cuda.synchronize()
    return C


def _f32(A, W, b, r, act, alpha):
    from wenet_amd import _lib
    L = _lib.lib()
    M, K = A.shape
    N = W.shape[0]
    C = torch.empty((M, N), device='cuda')
    _lib.check(L.wn_op_gemm(A.data_ptr(), W.data_ptr(), b.data_ptr() if b is not None else None,
                            r.data_ptr() if r is not None else None, C.data_ptr(), M, N, K,
                            alpha, act, torch.cuda.current_stream().cuda_stream), 'gemm')
    torch.cuda.synchronize()
    return C


def _ref(A, W, b, r, act, alpha):
    h = A.double() @ W.double().T
    if b is not None:
        h = h + b.double()
    h = {0: lambda t: t, 1: torch.nn.functional.silu, 2: torch.relu,
         3: torch.nn.functional.gelu}[act](h)
    h = alpha * h
    if r is not None:
        h = h + r.double()
    return h


@pytest.mark.parametrize('M,N,K,act,resid,bm', [
    (7932, 2048, 256, 1, False, 0),     # FFN w_1 of BASELINE config 2
    (7932, 256, 2048, 0, True, 0),      # FFN w_2 (one N tile: 128-row blocks)
    (7932, 768, 256, 0, False, 256),
    (1000, 4236, 256, 0, False, 128),   # ragged N (CTC-sized)
    (33, 20, 16, 2, True, 0),           # one k block, ragged everything
    (257, 516, 48, 3, False, 256),      # GELU, partial tiles on both sides
    (4096, 1024, 4864, 0, False, 0),    # long K (the subsampling output layer)
])
def test_gemm_x6_is_an_fp32_gemm(M, N, K, act, resid, bm):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = (torch.randn(N, generator=g) * 0.3).cuda()
    r = torch.randn(M, N, generator=g).cuda() if resid else None
    ref = _ref(A, W, b, r, act, 0.5)
    c6 = _x6(A, W, b, r, act, 0.5, bm)
    e6 = (c6.double() - ref).abs().max().item()
    r6 = ((c6.double() - ref) ** 2).mean().sqrt().item()
    assert torch.equal(c6, _x6(A, W, b, r, act, 0.5, bm))      # race screen
    if K % 32 != 0:          # the v_mfma_f32 kernel wants K % 32 == 0: torch fp32 instead
        assert e6 < 2e-6
        return
    c32 = _f32(A, W, b, r, act, 0.5)
    e32 = (c32.double() - ref).abs().max().item()
    r32 = ((c32.double() - ref) ** 2).mean().sqrt().item()
    print(f'\n[{M}x{N}x{K}] max |err| x6 {e6:.2e} / f32 mfma {e32:.2e}; rms {r6:.2e} / {r32:.2e}')
    assert e6 < 2e-5
    # an fp32 GEMM: not worse than the fp32 matrix-core kernel (noise margin 1.5x on the
    # maximum, 1.2x on the rms)
    assert e6 <= 1.5 * e32 + 1e-7
    assert r6 <= 1.2 * r32 + 1e-8


@pytest.mark.parametrize('M,N,K,act,bm', [
    (7932, 2048, 256, 1, 129), (7932, 2048, 256, 1, 130), (1000, 516, 48, 3, 130),
    (333, 256, 2048, 0, 129),
])
def test_gemm_x6_four_wave_tiles(M, N, K, act, bm):
    """128-row tiles on four waves, two blocks per CU (bm 129; 130: the lower half of the
    grid at s_setprio 3): same result as the 8-wave tiles, bit for bit (same products, same
    order per accumulator)."""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = (torch.randn(N, generator=g) * 0.3).cuda()
    ref = _x6(A, W, b, None, act, 1.0, 120)      # the 8-wave form of the 128-row tile
    got = _x6(A, W, b, None, act, 1.0, bm)
    assert torch.equal(ref, got)
    assert torch.equal(got, _x6(A, W, b, None, act, 1.0, bm))


def test_gemm_x6_vs_the_oracle_restatement():
    """The HIP kernel against oracle.x6_matmul (the same six plane products, summed in
    fp64): what separates them is only the fp32 accumulation order."""
    from oracle import wenet_oracle as O
    g = torch.Generator().manual_seed(17)
    A = torch.randn(777, 384, generator=g)
    W = torch.randn(300, 384, generator=g) / 384 ** 0.5
    ref = O.x6_matmul(A, W)
    c6 = _x6(A.cuda(), W.cuda(), None, None, 0, 1.0).cpu().double()
    # fp32 accumulation of 384 terms of magnitude ~1: a few ulp of the result
    assert (c6 - ref).abs().max().item() < 1e-5


def test_x6_planes_are_exact():
    """The operand split is exact (x0 + x1 + x2 == x for every fp32 input), so with W = I
    the GEMM must return A bit for bit up to the six-product rule: x * 1 keeps x0, x1, x2
    times the single plane of 1.0 -- all three survive (a0 b0, a1 b0, a2 b0)."""
    g = torch.Generator().manual_seed(5)
    A = (torch.randn(512, 256, generator=g) * torch.logspace(-20, 20, 256)).cuda()
    W = torch.eye(256).cuda()
    C = _x6(A, W, None, None, 0, 1.0)
    assert torch.equal(C, A)


@pytest.mark.parametrize('spread', ['operands', 'products'])
def test_gemm_x6_wide_dynamic_range_inside_one_dot_product(spread):
    """Operands whose magnitudes span 12 decades ALONG K (per-column logspace scale), so that
    one dot product mixes plane products of very different exponents: `operands` -- A scaled
    by s_k, W by 1 / s_k (products of comparable size from operands of wildly different
    size); `products` -- only A scaled (the terms of the sum themselves span 12 decades).
    Against fp64, normalised by sum_k |a_k w_k| (the natural scale of the rounding error of a
    dot product): the six-product GEMM is not worse than the v_mfma_f32 kernel and stays in
    the fp32 class (a few 2^-24 of the normaliser)."""
    M, N, K = 1024, 512, 512
    g = torch.Generator().manual_seed(31)
    sk = torch.logspace(-6, 6, K)[torch.randperm(K, generator=g)]
    A = torch.randn(M, K, generator=g) * sk
    W = torch.randn(N, K, generator=g) / (sk if spread == 'operands' else 1.0)
    ref = A.double() @ W.double().T
    norm = A.double().abs() @ W.double().abs().T
    c6 = _x6(A.cuda(), W.cuda(), None, None, 0, 1.0).cpu().double()
    c32 = _f32(A.cuda(), W.cuda(), None, None, 0, 1.0).cpu().double()
    e6 = ((c6 - ref).abs() / norm).max().item()
    e32 = ((c32 - ref).abs() / norm).max().item()
    r6 = (((c6 - ref) / norm) ** 2).mean().sqrt().item()
    r32 = (((c32 - ref) / norm) ** 2).mean().sqrt().item()
    print(f'\n[{spread}] max |err| / sum|a w|: x6 {e6:.2e} / f32 mfma {e32:.2e}; '
          f'rms {r6:.2e} / {r32:.2e}')
    assert e6 < 1e-6        # fp32 class: ~ sqrt(K) 2^-24 of the normaliser
    assert e6 <= 1.5 * e32 + 1e-9 and r6 <= 1.2 * r32 + 1e-10


@pytest.fixture(params=[0, 2], ids=['gemm_pair', 'hidden_on_chip'])
def ffn_form(request):
    """The feed-forward module as two six-product GEMMs (hidden tensor as a plane image in
    HBM) and as the fused kernel of csrc/ffn_x6f.hip (hidden tensor in registers; forced on
    every shape it supports: d_model 256, SiLU / ReLU)."""
    from wenet_amd import _lib
    L = _lib.lib()
    _lib.check(L.wn_tune_set(b'ffn_x6f', request.param), 'tune')
    yield request.param
    L.wn_tune_set(b'ffn_x6f', 1)


@pytest.mark.parametrize('M,D,F,act', [
    (128, 256, 128, 1), (700, 256, 2048, 3), (7932, 256, 2048, 1), (1000, 512, 2048, 2),
    (16231, 512, 2048, 1), (7932, 256, 2048, 2), (33, 256, 64, 1), (3000, 256, 1024, 1),
    (130, 256, 2048, 2), (12000, 256, 2048, 1),     # (12000 rows: two hidden slices)
    (295, 256, 2048, 1),      # one 11.8-s utterance: 3 row tiles x 32 hidden slices (round 5)
])
def test_ffn_x6_vs_fp64(M, D, F, act, ffn_form):
    from wenet_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(M + D + F + act)
    X = torch.randn(M, D, generator=g)
    W1 = torch.randn(F, D, generator=g) / D ** 0.5
    b1 = torch.randn(F, generator=g) * 0.3
    W2 = torch.randn(D, F, generator=g) / F ** 0.5
    b2 = torch.randn(D, generator=g) * 0.3
    x = torch.randn(M, D, generator=g)
    lw = 1.0 + 0.2 * torch.randn(D, generator=g)
    lb = 0.1 * torch.randn(D, generator=g)
    h = X.double() @ W1.double().T + b1.double()
    h = {1: torch.nn.functional.silu, 2: torch.relu, 3: torch.nn.functional.gelu}[act](h)
    xr = x.double() + 0.5 * (h @ W2.double().T + b2.double())
    yr = torch.nn.functional.layer_norm(xr, (D, ), lw.double(), lb.double(), 1e-5)
    t = [t.cuda().contiguous() for t in (X, W1, b1, W2, b2, lw, lb)]
    outs = []
    for _ in range(2):
        xo = x.clone().cuda()
        y = torch.empty((M, D), device='cuda')
        _lib.check(L.wn_op_ffn_x6(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(),
                                  t[3].data_ptr(), t[4].data_ptr(), xo.data_ptr(),
                                  t[5].data_ptr(), t[6].data_ptr(), y.data_ptr(), M, D, F, act,
                                  0.5, 1e-5, 1, torch.cuda.current_stream().cuda_stream),
                   'ffn_x6')
        torch.cuda.synchronize()
        outs.append((xo.cpu(), y.cpu()))
    (xo, y), (xo2, y2) = outs
    ex, ey = (xo.double() - xr).abs().max().item(), (y.double() - yr).abs().max().item()
    print(f'\n[{M} {D} {F}] max |err| x {ex:.2e} y {ey:.2e}')
    assert ex < 2e-5 and ey < 2e-5
    assert torch.equal(xo, xo2) and torch.equal(y, y2)


@pytest.mark.parametrize('config,B,frames,chunk', [('aishell_u2pp', 4, (400, 700), -1),
                                                   ('aishell_u2pp', 32, (800, 1200), -1),
                                                   ('wenetspeech_u2pp', 3, (300, 500), 16)])
def test_encoder_with_x6_ffn_matches_the_f32_mfma_path(config, B, frames, chunk):
    """Whole encoder with the feed-forward GEMMs on the six-product kernel (forced on small
    batches too) against every GEMM on v_mfma_f32: fp32 reordering noise only."""
    from gpu_util import cached_model
    from wenet_amd import _lib, synthetic as S
    L = _lib.lib()
    configs, sd, model = cached_model(config, 0)
    feats, lens = S.make_features(B, frames, seed=79)
    try:
        _lib.check(L.wn_tune_set(b'gemm_x6', 0), 'tune')
        ref, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        ref = ref.cpu()
        _lib.check(L.wn_tune_set(b'gemm_x6', 2), 'tune')
        got, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        got2, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        got = got.cpu()
    finally:
        L.wn_tune_set(b'gemm_x6', 1)
    assert torch.equal(got, got2.cpu())
    err = (got - ref).abs().max().item()
    print(f'\n[{config} B={B}] x6 FFN vs f32 MFMA: max |d enc| {err:.2e}')
    assert 0 < err < 1e-4


@pytest.mark.parametrize('config,B,frames,chunk', [('aishell_u2pp', 4, (400, 700), -1),
                                                   ('aishell_u2pp', 32, (800, 1200), -1),
                                                   ('aishell_u2pp', 3, (607, 911), 16),
                                                   ('aishell_u2pp', 1, (1181, 1181), -1),
                                                   # d_model 512: the GEMM pair takes the image
                                                   # (no x6_split launch); 32- and 64-row blocks
                                                   ('wenetspeech_u2pp', 5, (500, 830), 16),
                                                   ('librispeech_bidecoder_large', 50,
                                                    (900, 1100), -1)])
def test_ffn_input_as_a_plane_image_is_bit_identical(config, B, frames, chunk):
    """Round 5: the producers of LN(x) in front of a fused feed-forward module (the row-block
    pointwise_conv2 GEMM's LayerNorm epilogue, the partial reduction in front of the next layer's
    macaron module) write its X3 plane image and the fused kernel loads operand fragments, instead
    of fp32 rows that each of the S hidden-slice blocks loads, turns and splits (tune ffn_ximg).
    The split is exact and done on the same fp32 values, so the whole encoder must return the
    SAME BITS either way -- a wrong record address, a wrong half-wave exchange or a ragged last
    row tile shows up here.  (Row counts: ragged tiles of 32 and of 8 rows.)"""
    from gpu_util import cached_model
    from wenet_amd import _lib, synthetic as S
    L = _lib.lib()
    configs, sd, model = cached_model(config, 0)
    feats, lens = S.make_features(B, frames, seed=83, feat_dim=configs['input_dim'])
    try:
        _lib.check(L.wn_tune_set(b'ffn_ximg', 0), 'tune')
        ref, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        ref = ref.cpu()
        _lib.check(L.wn_tune_set(b'ffn_ximg', 1), 'tune')
        got, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        got2, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
    finally:
        L.wn_tune_set(b'ffn_ximg', 1)
    assert torch.isfinite(ref).all()
    assert torch.equal(got.cpu(), ref) and torch.equal(got2.cpu(), ref)


@pytest.mark.parametrize('config,B,frames,chunk', [('aishell_u2pp', 32, (800, 1200), -1),
                                                   ('aishell_u2pp', 24, (900, 1500), -1),
                                                   ('wenetspeech_u2pp', 20, (700, 1100), 16)])
def test_conv2_tile_forms_agree(config, B, frames, chunk):
    """tune x6_conv_bm (round 5): the subsampling conv2 as ONE launch of 128-row tiles on four
    waves (the default: it loses least to the prefix beam search it shares the chip with when
    decodes are in flight) against 256-row tiles on eight waves with the last round as K slices.
    Every output element of a full tile is accumulated over the same k blocks in the same order
    by both tile shapes; only the rows of the K-sliced last round are summed in another order
    (N <= 256) -- reordering noise at most, and nothing at all where that tail does not exist."""
    from gpu_util import cached_model
    from wenet_amd import _lib, synthetic as S
    L = _lib.lib()
    configs, sd, model = cached_model(config, 0)
    feats, lens = S.make_features(B, frames, seed=91)
    try:
        _lib.check(L.wn_tune_set(b'x6_conv_bm', 0), 'tune')
        ref, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        _lib.check(L.wn_tune_set(b'x6_conv_bm', 128), 'tune')
        got, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        got2, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
    finally:
        L.wn_tune_set(b'x6_conv_bm', 128)
    assert torch.isfinite(ref).all() and torch.equal(got.cpu(), got2.cpu())
    err = (got - ref).abs().max().item()
    print(f'\n[{config} B={B}] conv2 128-row one launch vs 256-row + K-slice tail: max |d enc| {err:.2e}')
    if configs['encoder_conf']['output_size'] > 256:
        assert err == 0.0       # no K-slice tail at N = 512: the same bits
    else:
        assert err < 2e-5


@pytest.mark.parametrize('ring', [3, 4, 5])
def test_ffn_on_chip_equals_itself_for_every_ring_depth(ring):
    """The DMA ring depth changes only when operands arrive, never what is multiplied: the
    fused module must return the same bits for three stages of 48 records (ring 3, the default)
    and four / five / six stages of 24 (a stale or early-overwritten
    stage shows up here), on a row count that leaves a ragged last tile."""
    from wenet_amd import _lib
    L = _lib.lib()
    M, D, F = 7932, 256, 2048
    g = torch.Generator().manual_seed(99)
    X = torch.randn(M, D, generator=g).cuda()
    W1 = (torch.randn(F, D, generator=g) / D ** 0.5).cuda()
    b1 = (torch.randn(F, generator=g) * 0.3).cuda()
    W2 = (torch.randn(D, F, generator=g) / F ** 0.5).cuda()
    b2 = (torch.randn(D, generator=g) * 0.3).cuda()
    x0 = torch.randn(M, D, generator=g)
    lw, lb = torch.ones(D).cuda(), torch.zeros(D).cuda()

    def run():
        xo = x0.clone().cuda()
        y = torch.empty((M, D), device='cuda')
        _lib.check(L.wn_op_ffn_x6(X.data_ptr(), W1.data_ptr(), b1.data_ptr(), W2.data_ptr(),
                                  b2.data_ptr(), xo.data_ptr(), lw.data_ptr(), lb.data_ptr(),
                                  y.data_ptr(), M, D, F, 1, 0.5, 1e-5, 3,
                                  torch.cuda.current_stream().cuda_stream), 'ffn_x6')
        torch.cuda.synchronize()
        return xo.cpu()
    # the stages of 24 records (rings 4..6) are compiled in WN_ABLATION builds only; the default
    # build runs the ring-3 leg as a determinism screen (3 runs, same bits)
    ablation = L.wn_tune_set(b'ffn_x6f_ring', 6) == 0
    if not ablation and ring != 3:
        pytest.skip('rings 4..6: WN_ABLATION build only')
    try:
        _lib.check(L.wn_tune_set(b'ffn_x6f', 2), 'tune')
        ref = run()
        if ablation:
            _lib.check(L.wn_tune_set(b'ffn_x6f_ring', ring), 'tune')
        for _ in range(3):
            assert torch.equal(run(), ref)
    finally:
        L.wn_tune_set(b'ffn_x6f', 1)
        if ablation:
            L.wn_tune_set(b'ffn_x6f_ring', 3)


def test_encoder_with_the_on_chip_ffn_matches_the_gemm_pair():
    """BASELINE config 2 shaped batch through the encoder with the fused six-product FFN
    (default at this size) against the two-GEMM form: same plane products, another fp32
    summation order only."""
    from gpu_util import cached_model
    from wenet_amd import _lib, synthetic as S
    L = _lib.lib()
    configs, sd, model = cached_model('aishell_u2pp', 0)
    feats, lens = S.make_features(32, (800, 1200), seed=80)
    try:
        _lib.check(L.wn_tune_set(b'x6_af32', 0), 'tune')     # (plane images: the fused form's input)
        _lib.check(L.wn_tune_set(b'ffn_x6f', 0), 'tune')
        ref, _ = model._forward_encoder(feats.cuda(), lens, -1, -1)
        ref = ref.cpu()
        _lib.check(L.wn_tune_set(b'ffn_x6f', 1), 'tune')
        got, _ = model._forward_encoder(feats.cuda(), lens, -1, -1)
        got2, _ = model._forward_encoder(feats.cuda(), lens, -1, -1)
    finally:
        L.wn_tune_set(b'ffn_x6f', 1)
    assert torch.equal(got, got2)
    err = (got.cpu() - ref).abs().max().item()
    print(f'\n[aishell B=32] on-chip FFN vs GEMM pair: max |d enc| {err:.2e}')
    assert 0 < err < 1e-4
