"""The fused fp32 feed-forward module (csrc/ffn_fused.hip) through the C ABI against an
fp64 torch evaluation of PositionwiseFeedForward + residual + LayerNorm
(positionwise_feed_forward.py:50-58, encoder_layer.py:220-228), and through the model
against the two-GEMM path it replaces."""
import numpy as np
import os

import pytest
import torch

from gpu_util import cached_model

pytestmark = pytest.mark.gpu


def _run(X, W1, b1, W2, b2, x, lw, lb, act, alpha):
    from wenet_amd import _lib
    L = _lib.lib()
    M, D = X.shape
    F = W1.shape[0]
    xo = x.clone().cuda()
    y = torch.empty((M, D), device='cuda')
    t = [t.cuda().contiguous() for t in (X, W1, b1, W2, b2, lw, lb)]
    _lib.check(L.wn_op_ffn_fused(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(),
                                 t[3].data_ptr(), t[4].data_ptr(), xo.data_ptr(),
                                 t[5].data_ptr(), t[6].data_ptr(), y.data_ptr(), M, D, F,
                                 act, alpha, 1e-5,
                                 torch.cuda.current_stream().cuda_stream), 'ffn_fused')
    torch.cuda.synchronize()
    return xo.cpu(), y.cpu()


@pytest.mark.parametrize('M,D,F,act', [
    (128, 256, 128, 1),       # 128-row blocks: one block, two chunks
    (64, 256, 64, 2),         # one block, one chunk, ReLU
    (700, 256, 2048, 1),      # ragged M, S = 16
    (700, 256, 2048, 3),      # ... GELU
    (7932, 256, 2048, 1),     # BASELINE config 2: 62 x 4 blocks
    (33000, 256, 2048, 1),    # more blocks than fit at once; S = 1 -> 32 chunks per block
    (1000, 512, 2048, 2),     # d = 512 (two W2 stages per k tile), ReLU
    (333, 512, 1024, 3),      # GELU
    (16231, 512, 2048, 1),    # BASELINE config 3: 127 x 2 blocks, 16 chunks per block
])
def test_ffn_fused_vs_fp64(M, D, F, act):
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
    xo, y = _run(X, W1, b1, W2, b2, x, lw, lb, act, 0.5)
    assert (xo.double() - xr).abs().max().item() < 2e-5
    assert (y.double() - yr).abs().max().item() < 2e-5
    xo2, y2 = _run(X, W1, b1, W2, b2, x, lw, lb, act, 0.5)   # race screen
    assert torch.equal(xo, xo2) and torch.equal(y, y2)


@pytest.mark.parametrize('config,B,frames,chunk', [('aishell_u2pp', 4, (400, 700), -1),
                                                   ('wenetspeech_u2pp', 3, (300, 500), 16)])
def test_encoder_with_fused_ffn_matches_the_gemm_pair(config, B, frames, chunk):
    """Whole encoder, fused FFN forced on a small batch (S = 16) against the two-GEMM
    path: same arithmetic up to the split of the hidden sum."""
    from wenet_amd import _lib, synthetic as S
    L = _lib.lib()
    configs, sd, model = cached_model(config, 0)
    feats, lens = S.make_features(B, frames, seed=71)
    try:
        _lib.check(L.wn_tune_set(b'ffn_fused', 0), 'tune')
        ref, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        ref = ref.cpu()
        _lib.check(L.wn_tune_set(b'ffn_fused', 2), 'tune')
        got, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        got = got.cpu()
    finally:
        L.wn_tune_set(b'ffn_fused', 1)
    err = (got - ref).abs().max().item()
    print(f'\n[{config}] fused vs GEMM pair: max |d enc| {err:.2e}')
    assert 0 < err < 2e-4 or err == 0.0


@pytest.mark.parametrize('B,frames,chunk', [(4, (400, 700), -1), (32, (800, 1200), -1),
                                            (3, (7, 90), 16)])
def test_encoder_with_rowln_gemm_matches_gemm_plus_layernorm(B, frames, chunk):
    """csrc/gemm_rowln.hip (out-projection / pointwise_conv2 + residual + the following
    LayerNorm in one launch, blocks own complete rows) against the GEMM + LayerNorm
    launches it replaces: whole AIShell encoder, ragged batches incl. the bench shape and
    utterances of one or two frames."""
    from wenet_amd import _lib, synthetic as S
    L = _lib.lib()
    configs, sd, model = cached_model('aishell_u2pp', 0)
    feats, lens = S.make_features(B, frames, seed=73)
    try:
        _lib.check(L.wn_tune_set(b'x6r', 0), 'tune')       # (the v_mfma_f32 forms under test)
        _lib.check(L.wn_tune_set(b'gemm_rowln', 0), 'tune')
        ref, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        ref = ref.cpu()
        _lib.check(L.wn_tune_set(b'gemm_rowln', 1), 'tune')
        got, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        got2, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        got = got.cpu()
    finally:
        L.wn_tune_set(b'gemm_rowln', 1)
        L.wn_tune_set(b'x6r', 1)
    assert torch.equal(got, got2.cpu())            # race screen
    err = (got - ref).abs().max().item()
    print(f'\n[B={B}] row-LN GEMM vs GEMM + LayerNorm: max |d enc| {err:.2e}')
    assert err < 2e-4


@pytest.mark.parametrize('config,B,frames,chunk', [('aishell_u2pp', 4, (400, 700), -1),
                                                   ('aishell_u2pp', 32, (800, 1200), -1),
                                                   ('wenetspeech_u2pp', 3, (300, 500), 16),
                                                   ('aishell_u2pp', 3, (7, 90), 16)])
def test_encoder_with_folded_relpos_attention_matches_the_two_contraction_form(config, B, frames,
                                                                               chunk):
    """(q + u).k + (q + v).p == q.(k + p) + (u.k + v.p): the rel-pos attention as ONE score
    contraction with rewritten keys and a per-key scalar (relpos_fold_kernel) against the
    two-contraction kernel, whole encoder, full-context and chunk-masked, ragged lengths."""
    from wenet_amd import _lib, synthetic as S
    L = _lib.lib()
    configs, sd, model = cached_model(config, 0)
    feats, lens = S.make_features(B, frames, seed=83)
    try:
        _lib.check(L.wn_tune_set(b'attn_fold', 0), 'tune')
        ref, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        ref = ref.cpu()
        _lib.check(L.wn_tune_set(b'attn_fold', 1), 'tune')
        got, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        got2, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        got = got.cpu()
        _lib.check(L.wn_tune_set(b'attn_fold', 2), 'tune')   # the fold as a separate pass
        sep, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        sep = sep.cpu()
        # the other issue order of the same arithmetic: one depthwise-conv output row per wave
        # (dwconv_tiled = 0; the kernel widths other than 256 / 512 run)
        _lib.check(L.wn_tune_set(b'attn_fold', 1), 'tune')
        _lib.check(L.wn_tune_set(b'dwconv_tiled', 0), 'tune')
        dwt, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        dwt = dwt.cpu()
    finally:
        L.wn_tune_set(b'attn_fold', 1)
        L.wn_tune_set(b'dwconv_tiled', 1)
    assert torch.equal(dwt, got), (dwt - got).abs().max().item()
    assert torch.equal(got, got2.cpu())            # race screen
    err = (got - ref).abs().max().item()
    err2 = (sep - ref).abs().max().item()
    print(f'\n[{config} B={B}] folded rel-pos attention vs two contractions: max |d enc| '
          f'{err:.2e} (in the kernel), {err2:.2e} (separate pass)')
    assert 0 < err < 1e-4 and 0 < err2 < 1e-4


@pytest.mark.parametrize('config,B,frames,chunk', [('aishell_u2pp', 4, (400, 700), -1),
                                                   ('aishell_u2pp', 32, (800, 1200), -1),
                                                   ('librispeech_bidecoder_large', 9, (900, 1310), -1),
                                                   ('wenetspeech_u2pp', 6, (600, 900), 16),
                                                   ('aishell_u2pp', 7, (130, 900), 16)])
def test_six_product_attention_matches_the_f32_mfma_attention(config, B, frames, chunk):
    """attention_x6.hip (both contractions of the folded rel-pos self attention as six bf16
    plane products of exactly split fp32 operands: key-tile images by a pack pass, then the
    kernel) against attention_kernel on v_mfma_f32: whole encoder, ragged lengths whose last key
    tile is partial, full context (the default route) and chunk masks (attn_x6 = 2).  The two do
    the same mathematics with other fp32 summation orders: 12 layers deep the encoder outputs
    differ by reordering noise only, and the six-product form is deterministic."""
    from wenet_amd import _lib, synthetic as S
    L = _lib.lib()
    configs, sd, model = cached_model(config, 0)
    feats, lens = S.make_features(B, frames, seed=87)
    try:
        _lib.check(L.wn_tune_set(b'attn_x6', 0), 'tune')
        ref, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        ref = ref.cpu()
        _lib.check(L.wn_tune_set(b'attn_x6', 2), 'tune')
        got, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        got2, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        got = got.cpu()
    finally:
        L.wn_tune_set(b'attn_x6', 1)
    assert torch.equal(got, got2.cpu())            # race screen
    err = (got - ref).abs().max().item()
    print(f'\n[{config} B={B} chunk {chunk}] six-product attention vs v_mfma_f32: max |d enc| '
          f'{err:.2e}')
    assert 0 < err < 5e-5


@pytest.mark.parametrize('M,N,epi', [(7932, 256, 1), (7932, 768, 0), (7932, 512, 0), (33, 256, 1),
                                     (1000, 256, 0), (31, 768, 0), (4097, 256, 1)])
def test_gemm_x6r_vs_fp64(M, N, epi):
    """csrc/gemm_x6r.hip (K = 256, A rows split in registers, W fragments straight from the
    plane image) against fp64: epi 0 plain projection, epi 1 residual + LayerNorm over complete
    rows; ragged row counts; deterministic."""
    from wenet_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(M + N + epi)
    A = torch.randn(M, 256, generator=g)
    W = torch.randn(N, 256, generator=g) / 16.0
    b = torch.randn(N, generator=g) * 0.3
    x = torch.randn(M, N, generator=g)
    lw = 1.0 + 0.2 * torch.randn(N, generator=g)
    lb = 0.1 * torch.randn(N, generator=g)
    ref = A.double() @ W.double().T + b.double()
    outs = []
    for _ in range(2):
        t = [v.cuda().contiguous() for v in (A, W, b, x.clone(), lw, lb)]
        y = torch.empty((M, N), device='cuda')
        C = torch.empty((M, N), device='cuda')
        _lib.check(L.wn_op_gemm_x6r(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(),
                                    t[3].data_ptr(), t[4].data_ptr(), t[5].data_ptr(),
                                    y.data_ptr(), C.data_ptr(), M, N, epi, 0.5, 1e-5, 1,
                                    torch.cuda.current_stream().cuda_stream), 'x6r')
        torch.cuda.synchronize()
        outs.append((C.cpu(), t[3].cpu(), y.cpu()))
    (C, xo, y), (C2, xo2, y2) = outs
    if epi == 0:
        err = (C.double() - ref).abs().max().item()
        print(f'\n[{M}x{N}] max |err| {err:.2e}')
        assert err < 5e-6 and torch.equal(C, C2)
    else:
        xr = x.double() + 0.5 * ref
        yr = torch.nn.functional.layer_norm(xr, (N, ), lw.double(), lb.double(), 1e-5)
        ex, ey = (xo.double() - xr).abs().max().item(), (y.double() - yr).abs().max().item()
        print(f'\n[{M}x{N}] max |err| x {ex:.2e} y {ey:.2e}')
        assert ex < 5e-6 and ey < 2e-5
        assert torch.equal(xo, xo2) and torch.equal(y, y2)


@pytest.mark.parametrize('M', [7932, 45, 1000])
def test_gemm_x6r_glu_vs_fp64(M):
    """csrc/gemm_x6r.hip epi 2: pointwise_conv1 + GLU (convolution.py:115-118) over the weight
    rows permuted per 64 as [32 values | 32 gates] (wn_model_create does that at load) against
    fp64; only the N / 2 GLU columns of C are written."""
    from wenet_amd import _lib
    L = _lib.lib()
    d = 256
    g = torch.Generator().manual_seed(M)
    A = torch.randn(M, d, generator=g)
    W = torch.randn(2 * d, d, generator=g) / 16.0       # reference order: [values | gates]
    b = torch.randn(2 * d, generator=g) * 0.3
    pre = A.double() @ W.double().T + b.double()
    ref = pre[:, :d] * torch.sigmoid(pre[:, d:])
    perm = torch.cat([torch.cat([torch.arange(32) + 32 * u, torch.arange(32) + 32 * u + d])
                      for u in range(d // 32)])
    Wp, bp = W[perm].contiguous(), b[perm].contiguous()
    outs = []
    for _ in range(2):
        t = [v.cuda().contiguous() for v in (A, Wp, bp)]
        C = torch.full((M, 2 * d), float('nan'), device='cuda')
        _lib.check(L.wn_op_gemm_x6r(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), None, None,
                                    None, None, C.data_ptr(), M, 2 * d, 2, 1.0, 1e-5, 1,
                                    torch.cuda.current_stream().cuda_stream), 'x6r')
        torch.cuda.synchronize()
        outs.append(C.cpu())
    err = (outs[0][:, :d].double() - ref).abs().max().item()
    print(f'\n[{M}] GLU max |err| {err:.2e}')
    assert err < 5e-6 and torch.equal(outs[0][:, :d], outs[1][:, :d])
    assert torch.isnan(outs[0][:, d:]).all()


@pytest.mark.parametrize('B,frames,chunk', [(32, (800, 1200), -1), (8, (300, 700), 16)])
def test_encoder_with_the_row_block_x6_gemms_matches_the_f32_forms(B, frames, chunk):
    """QKV, attention output projection + LayerNorm and pointwise_conv2 + LayerNorm on
    csrc/gemm_x6r.hip (default from 512 rows on) against the v_mfma_f32 forms: fp32
    reassociation noise only, deterministic."""
    from wenet_amd import _lib, synthetic as S
    L = _lib.lib()
    configs, sd, model = cached_model('aishell_u2pp', 0)
    feats, lens = S.make_features(B, frames, seed=74)
    try:
        _lib.check(L.wn_tune_set(b'x6r', 0), 'tune')
        ref, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        ref = ref.cpu()
        _lib.check(L.wn_tune_set(b'x6r', 1), 'tune')
        got, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        got2, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        got = got.cpu()
    finally:
        L.wn_tune_set(b'x6r', 1)
    assert torch.equal(got, got2.cpu())
    err = (got - ref).abs().max().item()
    print(f'\n[B={B}] row-block x6 GEMMs vs v_mfma_f32 forms: max |d enc| {err:.2e}')
    assert 0 < err < 2e-4


@pytest.mark.parametrize('B,frames', [(32, (800, 1200)), (5, (300, 1100))])
def test_chained_row_ln_glu_launch_is_bit_identical_to_the_two_launches(B, frames):
    """gemm_x6r.hip epi 3 (out-projection + residual + LN_conv chained with pointwise_conv1 + GLU,
    encoder_layer.py:236-251 / convolution.py:115-118, LN_conv(x) kept in LDS) runs the same
    arithmetic in the same order as epi 1 followed by epi 2: the encoder output is the same bits."""
    from wenet_amd import _lib, synthetic as S
    L = _lib.lib()
    configs, sd, model = cached_model('aishell_u2pp', 0)
    feats, lens = S.make_features(B, frames, seed=75)
    try:
        _lib.check(L.wn_tune_set(b'x6r_chain', 0), 'tune')
        ref, _ = model._forward_encoder(feats.cuda(), lens, -1, -1)
        ref = ref.cpu()
        _lib.check(L.wn_tune_set(b'x6r_chain', 1), 'tune')
        got, _ = model._forward_encoder(feats.cuda(), lens, -1, -1)
        got = got.cpu()
    finally:
        L.wn_tune_set(b'x6r_chain', 1)
    assert torch.isfinite(got).all()
    assert torch.equal(got, ref)


@pytest.mark.parametrize('B,frames,chunk', [(32, (800, 1200), -1), (5, (300, 1100), -1),
                                            (8, (300, 700), 16)])
def test_qkv_prologue_fold_is_bit_identical_to_the_reduce_launch(B, frames, chunk):
    """gemm_x6r.hip PRO (the QKV projection forms LN_mha(x + 0.5 FFN_macaron) itself from the
    slice partials of the fused feed-forward kernel, encoder_layer.py:220-232) does
    ffn_reduce_ln's arithmetic in the same order: the encoder output is the same bits as with
    the separate launch; ragged row counts (a last block with rows past M)."""
    from wenet_amd import _lib, synthetic as S
    L = _lib.lib()
    configs, sd, model = cached_model('aishell_u2pp', 0)
    feats, lens = S.make_features(B, frames, seed=76)
    try:
        _lib.check(L.wn_tune_set(b'x6r_pro', 0), 'tune')
        ref, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        ref = ref.cpu()
        _lib.check(L.wn_tune_set(b'x6r_pro', 1), 'tune')
        got, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        got2, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        got = got.cpu()
    finally:
        L.wn_tune_set(b'x6r_pro', 1)
    assert torch.isfinite(got).all()
    assert torch.equal(got, got2.cpu())
    assert torch.equal(got, ref), (got - ref).abs().max().item()


# ---- d_model = 512: the row-block kernels with the A image in LDS (csrc/gemm_x6r512.hip) ----------
@pytest.mark.parametrize('rows', [32, 64])
@pytest.mark.parametrize('M,N,epi', [(7932, 512, 1), (7932, 1536, 0), (16231, 512, 0), (33, 512, 1),
                                     (1000, 1024, 0), (31, 1536, 0), (4097, 512, 1),
                                     (7932, 512, 3), (45, 512, 3), (16231, 512, 3), (16231, 1536, 0),
                                     (65, 512, 3), (16231, 512, 1)])
def test_gemm_x6r512_vs_fp64(M, N, epi, rows):
    """K = 512: epi 0 plain projection (one to three 512-column passes over the same A image),
    epi 1 residual + LayerNorm over complete rows, epi 3 the same chained with pointwise_conv1 +
    GLU (weight rows permuted per 64 as [32 values | 32 gates]) from the LayerNorm rows in LDS;
    ragged row counts; deterministic.  Both block heights: 32 rows (A as a plane image in LDS)
    and 64 rows (fp32 rows in LDS, split per k block in registers; picked from M = 12288 on)."""
    from wenet_amd import _lib
    L = _lib.lib()
    _lib.check(L.wn_tune_set(b'x6r512_rows', rows), 'tune')
    try:
        _x6r512_case(M, N, epi)
    finally:
        L.wn_tune_set(b'x6r512_rows', 0)


def _x6r512_case(M, N, epi):
    from wenet_amd import _lib
    L = _lib.lib()
    d = 512
    g = torch.Generator().manual_seed(M + N + epi)
    A = torch.randn(M, d, generator=g)
    W = torch.randn(N, d, generator=g) / d ** 0.5
    b = torch.randn(N, generator=g) * 0.3
    x = torch.randn(M, N, generator=g)
    lw = 1.0 + 0.2 * torch.randn(N, generator=g)
    lb = 0.1 * torch.randn(N, generator=g)
    W2 = torch.randn(2 * d, d, generator=g) / d ** 0.5     # reference order: [values | gates]
    b2 = torch.randn(2 * d, generator=g) * 0.3
    perm = torch.cat([torch.cat([torch.arange(32) + 32 * u, torch.arange(32) + 32 * u + d])
                      for u in range(d // 32)])
    ref = A.double() @ W.double().T + b.double()
    outs = []
    for _ in range(2):
        t = [v.cuda().contiguous() for v in (A, W, b, x.clone(), lw, lb, W2[perm], b2[perm])]
        y = torch.full((M, N), float('nan'), device='cuda')
        C = torch.full((M, N), float('nan'), device='cuda')
        _lib.check(L.wn_op_gemm_x6r512(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(),
                                       t[3].data_ptr(), t[4].data_ptr(), t[5].data_ptr(),
                                       y.data_ptr(), t[6].data_ptr(), t[7].data_ptr(),
                                       C.data_ptr(), M, N, epi, 0.5, 1e-5, 1,
                                       torch.cuda.current_stream().cuda_stream), 'x6r512')
        torch.cuda.synchronize()
        outs.append((C.cpu(), t[3].cpu(), y.cpu()))
    (C, xo, y), (C2, xo2, y2) = outs
    if epi == 0:
        err = (C.double() - ref).abs().max().item()
        print(f'\n[{M}x{N}] max |err| {err:.2e}')
        assert err < 8e-6 and torch.equal(C, C2)
        return
    xr = x.double() + 0.5 * ref
    yr = torch.nn.functional.layer_norm(xr, (N, ), lw.double(), lb.double(), 1e-5)
    ex, ey = (xo.double() - xr).abs().max().item(), (y.double() - yr).abs().max().item()
    print(f'\n[{M}x{N}] max |err| x {ex:.2e} y {ey:.2e}')
    assert ex < 8e-6 and ey < 3e-5
    assert torch.equal(xo, xo2) and torch.equal(y, y2)
    if epi == 3:
        pre = yr @ W2.double().T + b2.double()
        cr = pre[:, :d] * torch.sigmoid(pre[:, d:])
        ec = (C.double() - cr).abs().max().item()
        print(f'[{M}x{N}] GLU max |err| {ec:.2e}')
        assert ec < 5e-5 and torch.equal(C, C2)


@pytest.mark.parametrize('config,B,frames,chunk', [('wenetspeech_u2pp', 32, (800, 1200), 16),
                                                   ('wenetspeech_u2pp', 5, (300, 1100), -1),
                                                   ('librispeech_bidecoder_large', 8, (300, 700), -1)])
def test_d512_row_block_kernels_match_the_other_forms(config, B, frames, chunk):
    """d_model = 512 encoders on the row-block kernels (gemm_x6r512.hip: QKV with the prologue
    fold, out-projection + LayerNorm + pointwise_conv1 + GLU chain, pointwise_conv2 + LayerNorm):
      * the prologue fold is ffn_reduce_ln's arithmetic in the same order: encoder output
        bit-identical to x6r_pro = 0 (the separate reduce launch);
      * the chain against the two launches (x6r_chain = 0): same products, same sums (the GLU
        GEMM then runs on the tile kernels): within fp32 rounding;
      * everything against the tile GEMMs + separate LayerNorms (x6r = 0): fp32 summation order."""
    from wenet_amd import _lib, synthetic as S
    L = _lib.lib()
    configs, sd, model = cached_model(config, 0)
    feats, lens = S.make_features(B, frames, seed=77)
    try:
        got, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        got2, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        got = got.cpu()
        _lib.check(L.wn_tune_set(b'x6r_pro', 0), 'tune')
        nopro, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        nopro = nopro.cpu()
        _lib.check(L.wn_tune_set(b'x6r_pro', 1), 'tune')
        _lib.check(L.wn_tune_set(b'x6r_chain', 0), 'tune')
        nochain, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        nochain = nochain.cpu()
        _lib.check(L.wn_tune_set(b'x6r_chain', 1), 'tune')
        _lib.check(L.wn_tune_set(b'x6r', 0), 'tune')
        ref, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        ref = ref.cpu()
    finally:
        L.wn_tune_set(b'x6r_pro', 1)
        L.wn_tune_set(b'x6r_chain', 1)
        L.wn_tune_set(b'x6r', 1)
    assert torch.isfinite(got).all()
    assert torch.equal(got, got2.cpu())                 # race screen
    assert torch.equal(got, nopro), (got - nopro).abs().max().item()
    e1, e2 = (got - nochain).abs().max().item(), (got - ref).abs().max().item()
    print(f'\n[{config} B={B}] d512 row-block kernels: vs two launches {e1:.2e}, vs tile GEMMs {e2:.2e}')
    assert e1 < 1e-4 and 0 < e2 < 2e-4


@pytest.mark.parametrize('config,B,frames,chunk', [('aishell_u2pp', 32, (800, 1200), -1),
                                                   ('aishell_u2pp', 7, (30, 900), 16),
                                                   ('aishell_conformer', 6, (300, 700), -1)])
def test_depthwise_conv_prologue_matches_the_separate_launch(config, B, frames, chunk):
    """gemm_x6r.hip DWC (the pointwise_conv2 row-block GEMM forms depthwise conv + LayerNorm /
    eval-BatchNorm + SiLU of the GLU output itself, convolution.py:119-148) does
    dwconv_tiled_kernel's operations in the same order per output row -- causal (K = 8, left pad
    frames) and symmetric (K = 15, BatchNorm affine) kernels, ragged utterances whose rows share
    8-row groups, very short utterances, chunk masks.  Not the same BITS: the two translation
    units compile the shared LayerNorm / SiLU source under different vectoriser settings and
    contract a multiply-add differently (measured 2.9e-6 on unit-scale encoder outputs, r06k;
    making the fma explicit moved every LayerNorm of the path by an ulp and one utterance in 256
    across a pruning tie, so the source stays as it was); deterministic, and within 2e-5."""
    from wenet_amd import _lib, synthetic as S
    L = _lib.lib()
    configs, sd, model = cached_model(config, 0)
    feats, lens = S.make_features(B, frames, seed=91)
    try:
        _lib.check(L.wn_tune_set(b'x6r_dwc', 0), 'tune')
        ref, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        ref = ref.cpu()
        _lib.check(L.wn_tune_set(b'x6r_dwc', 1), 'tune')
        got, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        got2, _ = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        got = got.cpu()
    finally:
        L.wn_tune_set(b'x6r_dwc', 1)
    assert torch.isfinite(got).all()
    assert torch.equal(got, got2.cpu())
    err = (got - ref).abs().max().item()
    print(f'\n[{config} B={B}] depthwise-conv prologue vs separate launch: max |d enc| {err:.2e}')
    assert err < 2e-5
