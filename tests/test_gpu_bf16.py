"""GPU parity of the bf16-operand mode (WN_PREC_BF16, `recognize.py --dtype bf16`,
BASELINE.json configs[4]): the bf16 matrix-core GEMM (csrc/gemm_bf16.hip) and the
decode path running on it, against the oracle with the SAME operand rounding
(oracle.wenet_oracle.bf16_operands: both operands of every GEMM-kernel
contraction rounded to bf16 RNE, fp32 accumulation, everything else fp32).

Tolerances:
  single contraction   |err| <= 2e-6 * (|A| @ |W|^T)   (fp32 accumulation order only)
  encoder layers       |err| <= 6e-3 * scale per layer (an activation that sits on a
                       bf16 rounding boundary may round the other way on the GPU:
                       one bf16 ulp = 0.4 % of that element; the attention kernel
                       rounds un-normalised online-softmax probabilities, the oracle
                       normalised ones)
  decode stages        fed identical inputs: CTC log-probs <= 5e-3, rescoring scores
                       <= 3e-2, n-best lists identical on the same log-probs
The bf16 mode is NOT the parity mode against the fp32 reference (that is the
default fp32 path); its distance from fp32 is reported, not bounded tightly.
"""
import numpy as np
import os

import pytest
import torch

from gpu_util import cached_model, compare_nbest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[0, 1], ids=['convert', 'stored'])
def bf16_store(request):
    """Every test runs in both forms of the bf16 mode: operands converted on the fly
    from fp32 tensors (0) and the bf16-STORAGE form (1: LayerNorm output, FFN hidden
    and attention context kept as bf16, bf16 image of the weight slab).  The two
    are the same arithmetic -- rounding is idempotent -- so the expectations are
    identical."""
    from wenet_amd import _lib
    L = _lib.lib()
    _lib.check(L.wn_tune_set(b'bf16_store', request.param), 'tune')
    yield request.param
    L.wn_tune_set(b'bf16_store', 1)  # the shipped default


def _oracle():
    from oracle import wenet_oracle as O
    return O


def _r(t):
    return t.to(torch.bfloat16).to(torch.float64)


def _ptr(t):
    return None if t is None else t.data_ptr()


def _gemm_bf16(A, W, bias=None, resid=None, alpha=1.0, act=0):
    from wenet_amd import _lib
    L = _lib.lib()
    M, K = A.shape
    N = W.shape[0]
    C = torch.empty((M, N), dtype=torch.float32, device='cuda')
    _lib.check(L.wn_op_gemm_bf16(_ptr(A), _ptr(W), _ptr(bias), _ptr(resid), _ptr(C),
                                 M, N, K, alpha, act,
                                 torch.cuda.current_stream().cuda_stream), 'gemm_bf16')
    torch.cuda.synchronize()
    return C


@pytest.mark.parametrize('M,N,K', [
    (128, 128, 32), (1, 1, 32), (77, 67, 64), (300, 200, 96), (300, 256, 256),
    (7936, 2048, 256), (7936, 256, 2048), (513, 4233, 256), (2000, 768, 256),
    (129, 130, 2432), (3000, 1280, 5120), (3000, 5120, 1280), (4097, 384, 160)])
def test_gemm_bf16_plain_asymmetric(M, N, K):
    """Both K-tile widths (K % 64 == 0 and not), every block shape, ragged M / N."""
    g = torch.Generator().manual_seed(M * 131 + N * 7 + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g)  # asymmetric operands (transpose check)
    ref = _r(A) @ _r(W).T
    got = _gemm_bf16(A.cuda(), W.cuda()).cpu().double()
    scale = (A.abs().double() @ W.abs().double().T).max().item()
    err = (got - ref).abs().max().item()
    assert err <= 2e-6 * scale, (err, scale)
    # ... and the operands really were rounded: the exact product is further away
    exact = A.double() @ W.double().T
    assert (got - exact).abs().max().item() > 5 * err + 1e-9 * scale


def test_gemm_bf16_identity_detects_transposed_store():
    K = 64
    A = torch.eye(K)
    W = torch.arange(96 * K, dtype=torch.float32).reshape(96, K) / 100.0
    got = _gemm_bf16(A.cuda(), W.cuda()).cpu()
    want = W.to(torch.bfloat16).to(torch.float32).T.contiguous()
    torch.testing.assert_close(got, want, rtol=0, atol=0)


def test_gemm_bf16_rounding_is_nearest_even():
    """Operand values half way between two bf16 numbers (the tie) and just off
    it: the kernel's conversion must agree with torch's (RNE) on every one."""
    base = torch.tensor([1.0, 1.5, 3.0, 100.0, 0.007]).to(torch.bfloat16).float()
    ulp = torch.tensor([2.0 ** -7, 2.0 ** -7, 2.0 ** -6, 2.0 ** -1, 2.0 ** -15])
    vals = []
    for j in range(8):
        lo = base + j * ulp                     # a bf16 number
        vals += [lo + ulp / 2, lo + ulp / 2 * 1.001, lo + ulp / 2 * 0.999]
    v = torch.cat(vals)
    v = torch.cat([v, -v])
    K = 32 * ((v.numel() + 31) // 32)
    A = torch.zeros(K, K)
    A[torch.arange(v.numel()), torch.arange(v.numel())] = v
    W = torch.eye(K)
    got = _gemm_bf16(A.cuda(), W.cuda()).cpu()
    want = A.to(torch.bfloat16).to(torch.float32)
    torch.testing.assert_close(got, want, rtol=0, atol=0)


@pytest.mark.parametrize('act', [0, 1, 2, 3])
@pytest.mark.parametrize('use_resid', [False, True])
@pytest.mark.parametrize('K', [96, 128])
def test_gemm_bf16_epilogues(act, use_resid, K):
    g = torch.Generator().manual_seed(act * 2 + int(use_resid) + K)
    M, N = 333, 200
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.2
    bias = torch.randn(N, generator=g)
    resid = torch.randn(M, N, generator=g) if use_resid else None
    y = (_r(A) @ _r(W).T).float() + bias
    if act == 1:
        y = torch.nn.functional.silu(y)
    elif act == 2:
        y = torch.relu(y)
    elif act == 3:
        y = torch.nn.functional.gelu(y)
    y = 0.5 * y
    if use_resid:
        y = y + resid
    got = _gemm_bf16(A.cuda(), W.cuda(), bias.cuda(),
                     resid.cuda() if use_resid else None, 0.5, act).cpu()
    torch.testing.assert_close(got, y, rtol=1e-5, atol=2e-5)


def _gemm_stored(A, W, bias=None, resid=None, alpha=1.0, act=0, c_bf16=False):
    from wenet_amd import _lib
    L = _lib.lib()
    M, K = A.shape
    N = W.shape[0]
    C = torch.empty((M, N), dtype=torch.bfloat16 if c_bf16 else torch.float32,
                    device='cuda')
    _lib.check(L.wn_op_gemm_bf16_stored(_ptr(A), _ptr(W), _ptr(bias), _ptr(resid), _ptr(C),
                                        M, N, K, alpha, act, int(c_bf16),
                                        torch.cuda.current_stream().cuda_stream),
               'gemm_bf16_stored')
    torch.cuda.synchronize()
    return C


@pytest.mark.parametrize('M,N,K', [
    (128, 128, 32), (1, 1, 32), (77, 67, 64), (300, 200, 96), (300, 256, 256),
    (7936, 2048, 256), (7936, 256, 2048), (513, 4233, 256), (129, 130, 2432),
    (3000, 1280, 5120), (6000, 5120, 1280), (4097, 384, 160), (2100, 1280, 128)])
def test_gemm_bf16_stored_plain(M, N, K, bf16_store):
    """The bf16-storage GEMM (A, W bf16 in HBM, prefetch distance 2): every block
    shape incl. 256x256, both K-tile widths, odd / even K-tile counts, ragged M / N;
    fp32 and bf16 C."""
    if bf16_store == 0:
        pytest.skip('one run is enough for the raw operator')
    g = torch.Generator().manual_seed(M * 131 + N * 7 + K + 1)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g)
    ref = _r(A) @ _r(W).T
    scale = (A.abs().double() @ W.abs().double().T).max().item()
    got = _gemm_stored(A.cuda(), W.cuda()).cpu().double()
    err = (got - ref).abs().max().item()
    assert err <= 2e-6 * scale, (err, scale)
    # identical to the convert-on-the-fly kernel up to accumulation order
    conv = _gemm_bf16(A.cuda(), W.cuda()).cpu().double()
    assert (got - conv).abs().max().item() <= 4e-6 * scale
    got16 = _gemm_stored(A.cuda(), W.cuda(), c_bf16=True).cpu()
    assert got16.dtype == torch.bfloat16
    want16 = ref.float().to(torch.bfloat16).float()
    # a value on a bf16 tie may round either way after an accumulation-order change
    assert (got16.float() - want16).abs().max().item() <= \
        2.0 ** -7 * ref.abs().max().item() + 4e-6 * scale


def _gemm_lowp_bf16(A16, W16, bias=None, resid=None, alpha=1.0, act=0, c_bf16=False):
    from wenet_amd import _lib
    L = _lib.lib()
    M, K = A16.shape
    N = W16.shape[0]
    C = torch.empty((M, N), dtype=torch.bfloat16 if c_bf16 else torch.float32, device='cuda')
    _lib.check(L.wn_op_gemm_lowp(_ptr(A16), _ptr(W16), None, None, _ptr(bias), _ptr(resid),
                                 _ptr(C), None, M, N, K, alpha, act, 1 if c_bf16 else 0, 1,
                                 torch.cuda.current_stream().cuda_stream), 'gemm_lowp')
    torch.cuda.synchronize()
    return C


@pytest.mark.parametrize('M,N,K,act,mode', [
    (512, 512, 128, 0, 'plain'),        # 2 K tiles: prologue + one pair
    (700, 520, 192, 1, 'plain'),        # ragged M and N, odd K-tile count
    (256, 256, 1280, 3, 'c_bf16'),
    (3000, 1280, 5120, 0, 'resid'),     # Whisper w_2
    (2900, 5120, 1280, 3, 'c_bf16'),    # Whisper w_1 (GELU, bf16 hidden)
    (1500, 3840, 1280, 0, 'plain'),     # Whisper QKV
    (1000, 264, 320, 2, 'resid'),       # N not a multiple of 32
])
def test_gemm_bf16_pipelined(M, N, K, act, mode, bf16_store):
    """gemm_bf16p_kernel (256x256, direct-to-LDS DMA, counted vmcnt, staggered waves,
    operand-swapped MFMA): against the fp64 product of the bf16 operands, bitwise equal
    to itself over repeated launches (race screen: the LDS ring is ordered only by the
    counted waits and barriers), and equal to the register-staged kernel up to
    accumulation order."""
    if bf16_store == 0:
        pytest.skip('one run is enough for the raw operator')
    from wenet_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K + act)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g) * 0.3).to(torch.bfloat16)
    bias = torch.randn(N, generator=g)
    resid = torch.randn(M, N, generator=g) if mode == 'resid' else None
    acc = A.double() @ W.double().T
    y = acc.float() + bias
    if act == 1:
        y = torch.nn.functional.silu(y)
    elif act == 2:
        y = torch.relu(y)
    elif act == 3:
        y = torch.nn.functional.gelu(y)
    y = 0.5 * y
    if resid is not None:
        y = y + resid
    scale = (A.abs().double() @ W.abs().double().T).max().item()
    args = (A.cuda(), W.cuda(), bias.cuda(), resid.cuda() if resid is not None else None,
            0.5, act, mode == 'c_bf16')
    try:
        _lib.check(L.wn_tune_set(b'gemm_tile_bf16', 8), 'tune')
        got = _gemm_lowp_bf16(*args)
        for _ in range(4):
            again = _gemm_lowp_bf16(*args)
            assert torch.equal(got, again), 'pipelined GEMM is not deterministic (race?)'
        _lib.check(L.wn_tune_set(b'gemm_tile_bf16', 1), 'tune')
        other = _gemm_lowp_bf16(*args)
    finally:
        L.wn_tune_set(b'gemm_tile_bf16', 0)
    got, other = got.cpu().float(), other.cpu().float()
    if mode == 'c_bf16':
        tol = 2.0 ** -7 * y.abs().max().item() + 4e-6 * scale
        assert (got - y).abs().max().item() <= tol
        assert (got - other).abs().max().item() <= tol
    else:
        assert (got - y).abs().max().item() <= 3e-6 * scale + 2e-5
        assert (got - other).abs().max().item() <= 4e-6 * scale + 2e-5


@pytest.mark.parametrize('act', [0, 1, 2, 3])
@pytest.mark.parametrize('mode', ['plain', 'resid', 'c_bf16'])
@pytest.mark.parametrize('K', [96, 128])
def test_gemm_bf16_stored_epilogues(act, mode, K, bf16_store):
    if bf16_store == 0:
        pytest.skip('one run is enough for the raw operator')
    g = torch.Generator().manual_seed(act * 3 + K + len(mode))
    M, N = 333, 200
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.2
    bias = torch.randn(N, generator=g)
    resid = torch.randn(M, N, generator=g) if mode == 'resid' else None
    y = (_r(A) @ _r(W).T).float() + bias
    if act == 1:
        y = torch.nn.functional.silu(y)
    elif act == 2:
        y = torch.relu(y)
    elif act == 3:
        y = torch.nn.functional.gelu(y)
    y = 0.5 * y
    if resid is not None:
        y = y + resid
    got = _gemm_stored(A.cuda(), W.cuda(), bias.cuda(),
                       resid.cuda() if resid is not None else None, 0.5, act,
                       c_bf16=(mode == 'c_bf16')).cpu()
    if mode == 'c_bf16':
        torch.testing.assert_close(got.float(), y, rtol=2.0 ** -7, atol=1e-3)
    else:
        torch.testing.assert_close(got, y, rtol=1e-5, atol=2e-5)


def _set_dtype(model, dtype):
    model.set_compute_dtype(dtype)
    assert model.compute_dtype == dtype


@pytest.mark.parametrize('config,B,frames,chunk,left', [
    ('tiny_causal', 6, (30, 260), -1, -1),
    ('tiny_causal', 4, (30, 200), 8, 1),
    ('tiny_sym', 5, (7, 180), -1, -1),
    ('tiny_bn', 4, (20, 150), -1, -1),
    ('whisper_tiny_like', 4, (3, 140), -1, -1),
    ('whisper_tiny_like', 2, (780, 900), -1, -1),     # 4 waves per attention block
    ('whisper_tiny_like', 2, (2050, 2600), -1, -1),   # 8 waves per attention block
])
def test_bf16_encoder_layers_vs_oracle(config, B, frames, chunk, left):
    """Every encoder layer output of the bf16 mode against the oracle with the same
    operand rounding (GLU, implicit-GEMM conv, gathered-row conv, residual and
    activation epilogues of the bf16 kernels all sit on this path)."""
    from wenet_amd import _lib, synthetic as S
    O = _oracle()
    configs, sd, model = cached_model(config, 0)
    feats, lens = S.make_features(B, frames, seed=91, feat_dim=configs['input_dim'])
    with torch.no_grad():
        ref32, _, layers32 = O.encoder_forward(configs, sd, feats, lens, chunk, left,
                                               return_layers=True)
        with O.bf16_operands(sd):
            ref, mask, layers = O.encoder_forward(configs, sd, feats, lens, chunk,
                                                  left, return_layers=True)
    ref_lens = mask.squeeze(1).sum(1).numpy()
    L = _lib.lib()
    _set_dtype(model, 'bf16')
    try:
        for n in range(len(layers)):
            _lib.check(L.wn_debug_set(model._h, b'n_layers', n), 'dbg')
            _lib.check(L.wn_debug_set(model._h, b'skip_after_norm', 1), 'dbg')
            enc, m = model._forward_encoder(feats.cuda(), lens, chunk, left)
            enc = enc.cpu()
            worst, apart = 0.0, 0.0
            for b in range(B):
                nb = int(ref_lens[b])
                if nb:
                    scale = max(layers[n][b, :nb].abs().max().item(), 1.0)
                    worst = max(worst, (enc[b, :nb] - layers[n][b, :nb]).abs().max().item() / scale)
                    apart = max(apart, (layers32[n][b, :nb] - layers[n][b, :nb]).abs().max().item() / scale)
            assert worst < 6e-3, (config, 'layer', n, worst)
            # the emulation and fp32 differ by far more than GPU vs emulation,
            # so this test does tell the two arithmetic modes apart
            if n == len(layers) - 1:
                assert apart > 1.5 * worst, (config, apart, worst)
    finally:
        L.wn_debug_set(model._h, b'n_layers', -1)
        L.wn_debug_set(model._h, b'skip_after_norm', 0)
        _set_dtype(model, 'fp32')


@pytest.mark.parametrize('config,B,frames,chunk', [
    ('tiny_causal', 6, (40, 260), -1), ('tiny_causal', 4, (40, 200), 8),
    ('tiny_sym', 5, (20, 180), -1), ('tiny_bn', 4, (30, 150), -1)])
def test_bf16_decode_stages_vs_oracle(config, B, frames, chunk):
    """The stages after the encoder in the bf16 mode, each fed the SAME inputs as
    the oracle under the same operand rounding (the sharpened CTC head multiplies
    an encoder difference by ~12, so end-to-end score comparisons would only
    measure the encoder tolerance again):
      CTC head       oracle encoder output -> log-probs, |err| <= 5e-3
      searches       the GPU's own log-probs -> oracle searches: identical n-best
      rescoring      oracle encoder output + oracle n-best -> scores within 3e-2
      end to end     greedy tokens identical wherever the top-1 margin > 0.2."""
    from wenet_amd import search as WS, synthetic as S
    O = _oracle()
    configs, sd, model = cached_model(config, 0)
    feats, lens = S.make_features(B, frames, seed=17, feat_dim=configs['input_dim'])
    rw = 0.3 if configs.get('decoder') == 'bitransformer' else 0.0
    sos, eos = O.special_symbols(configs)
    with torch.no_grad(), O.bf16_operands(sd):
        enc, mask = O.encoder_forward(configs, sd, feats, lens, chunk, -1)
        enc_lens = mask.squeeze(1).sum(1)
        logp = O.ctc_logprobs(sd, enc)
        pre = O.ctc_prefix_beam_search(logp, enc_lens, 4)
        ref_resc = O.attention_rescoring(configs, sd, pre, enc, enc_lens, 0.5, rw,
                                         sos, eos)
        ref_greedy = O.ctc_greedy_search(logp, enc_lens)
    top2 = logp.topk(2, dim=-1).values
    margin = top2[..., 0] - top2[..., 1]
    _set_dtype(model, 'bf16')
    try:
        # CTC head on the oracle's encoder output
        logp_g = model.ctc_logprobs(enc.cuda(), encoder_lens=enc_lens).cpu()
        for b in range(B):
            n = int(enc_lens[b])
            assert (logp_g[b, :n] - logp[b, :n]).abs().max().item() < 5e-3, (config, b)
        # rescoring on the oracle's encoder output and n-best lists
        got_resc = WS.attention_rescoring(model, pre, enc.cuda(), enc_lens, 0.5, rw)
        for b in range(B):
            np.testing.assert_allclose(got_resc[b].all_scores, ref_resc[b].all_scores,
                                       rtol=0, atol=3e-2)
        # searches on the GPU's own bf16-mode log-probs
        enc_g, mask_g = model._forward_encoder(feats.cuda(), lens, chunk, -1)
        lens_g = mask_g.squeeze(1).sum(1).cpu()
        assert lens_g.tolist() == enc_lens.tolist()
        own = model.ctc_logprobs(enc_g, encoder_lens=lens_g).cpu()
        want = O.ctc_prefix_beam_search(own, lens_g, 4)
        got = model.decode(['ctc_greedy_search', 'ctc_prefix_beam_search'],
                           feats.cuda(), lens, beam_size=4, decoding_chunk_size=chunk)
    finally:
        _set_dtype(model, 'fp32')
    for b in range(B):
        compare_nbest(got['ctc_prefix_beam_search'][b], want[b].nbest,
                      want[b].nbest_scores, want[b].nbest_times, what=f'{config}[{b}]')
        n = int(enc_lens[b])
        if n and margin[b, :n].min().item() > 0.2:
            assert got['ctc_greedy_search'][b].tokens == ref_greedy[b].tokens


@pytest.mark.parametrize('nw', [2, 4, 8])
@pytest.mark.parametrize('config,chunk,left', [('tiny_causal', 8, 1), ('tiny_causal', -1, -1),
                                               ('tiny_sym', -1, -1)])
def test_bf16_attention_block_shapes(nw, config, chunk, left):
    """Every block shape of the bf16 attention kernel (2 / 4 / 8 query groups per
    block) under the full, chunk and left-context masks, rel-pos variant, ragged
    lengths that leave partial query groups and partial key tiles."""
    from wenet_amd import _lib, synthetic as S
    O = _oracle()
    configs, sd, model = cached_model(config, 0)
    feats, lens = S.make_features(4, (300, 700), seed=29, feat_dim=configs['input_dim'])
    with torch.no_grad(), O.bf16_operands(sd):
        ref, mask = O.encoder_forward(configs, sd, feats, lens, chunk, left)
    ref_lens = mask.squeeze(1).sum(1).numpy()
    L = _lib.lib()
    _set_dtype(model, 'bf16')
    try:
        _lib.check(L.wn_tune_set(b'attn_bf16_nw', nw), 'tune')
        enc, _ = model._forward_encoder(feats.cuda(), lens, chunk, left)
    finally:
        L.wn_tune_set(b'attn_bf16_nw', 0)
        _set_dtype(model, 'fp32')
    enc = enc.cpu()
    for b in range(4):
        nb = int(ref_lens[b])
        scale = max(ref[b, :nb].abs().max().item(), 1.0)
        err = (enc[b, :nb] - ref[b, :nb]).abs().max().item() / scale
        assert err < 6e-3, (config, nw, b, err)


@pytest.mark.parametrize('B,frames,seed,nw', [
    (2, (780, 900), 5, 0),       # 4 query groups per block (the default at every length)
    (3, (900, 2600), 6, 8),      # 8 per block, ragged: short sequences end stages early
    (2, (2050, 2600), 7, 0),
    (4, (1290, 1300), 8, 0),     # lengths reset to multiples of 64 keys and one past (below)
])
def test_bf16_attention_dma_staging_is_bit_identical(B, frames, seed, nw):
    """The LDS-DMA staged bf16 self attention (K and V rows of the QKV matrix, PV fragments
    through transpose reads, attention_bf16_dma_kernel) does the register-staged kernel's
    arithmetic in the same order: the encoder outputs of the two are the same bits, on ragged batches whose
    last stage is partial; and both stay within the oracle's bf16 emulation."""
    from wenet_amd import _lib, synthetic as S
    O = _oracle()
    configs, sd, model = cached_model('whisper_tiny_like', 0)
    feats, lens = S.make_features(B, frames, seed=seed, feat_dim=configs['input_dim'])
    if B == 4:   # exact multiples of 128 frames = 64 encoder frames, and one past
        lens = torch.tensor([1280, 1282, 1152, 770], dtype=torch.int32)
        feats = feats[:, :1282].clone()
        for b in range(B):
            feats[b, int(lens[b]):] = 0.0      # zero padding, as a collated batch has it
    L = _lib.lib()
    _set_dtype(model, 'bf16')
    try:
        _lib.check(L.wn_tune_set(b'attn_bf16_nw', nw), 'tune')
        _lib.check(L.wn_tune_set(b'attn_bf16_defer', 0), 'tune')   # rescale whenever a maximum moves
        _lib.check(L.wn_tune_set(b'attn_bf16_dma', 1), 'tune')   # the LDS-DMA staged kernel
        enc1, _ = model._forward_encoder(feats.cuda(), lens)
        enc1b, _ = model._forward_encoder(feats.cuda(), lens)
        # deferred rescale (the default, threshold 8 in log2 units) and a threshold that makes the
        # update branch fire in mid-sequence tiles: other roundings of P, the same softmax
        deferred = []
        for thr10 in (80, 5, 20):
            _lib.check(L.wn_tune_set(b'attn_bf16_defer', thr10), 'tune')
            e, _ = model._forward_encoder(feats.cuda(), lens)
            deferred.append(e.cpu())
        _lib.check(L.wn_tune_set(b'attn_bf16_dma', 0), 'tune')
        enc0, _ = model._forward_encoder(feats.cuda(), lens)
    finally:
        L.wn_tune_set(b'attn_bf16_dma', 1)      # the defaults
        L.wn_tune_set(b'attn_bf16_defer', 80)
        L.wn_tune_set(b'attn_bf16_nw', 0)
        _set_dtype(model, 'fp32')
    assert torch.equal(enc1, enc1b), 'DMA-staged attention is not deterministic (race?)'
    assert torch.equal(enc1, enc0), (enc1 - enc0).abs().max().item()
    with torch.no_grad(), O.bf16_operands(sd):
        ref, mask = O.encoder_forward(configs, sd, feats, lens, -1, -1)
    ref_lens = mask.squeeze(1).sum(1).numpy()
    enc1 = enc1.cpu()
    for b in range(B):
        nb = int(ref_lens[b])
        scale = max(ref[b, :nb].abs().max().item(), 1.0)
        assert (enc1[b, :nb] - ref[b, :nb]).abs().max().item() / scale < 6e-3
        for e in deferred:
            assert torch.isfinite(e[b, :nb]).all()
            assert (e[b, :nb] - ref[b, :nb]).abs().max().item() / scale < 6e-3
            assert (e[b, :nb] - enc1[b, :nb]).abs().max().item() / scale < 4e-3


def test_bf16_is_a_per_handle_switch_and_fp32_comes_back_bit_exact():
    """fp32 -> bf16 -> fp32 on one handle: the two fp32 runs are bit-identical, the
    bf16 run is not; clones inherit the mode of their source at clone time."""
    from wenet_amd import synthetic as S
    configs, sd, model = cached_model('tiny_causal', 0)
    feats, lens = S.make_features(3, (50, 120), seed=3, feat_dim=configs['input_dim'])
    a, _ = model._forward_encoder(feats.cuda(), lens)
    a = a.clone()
    _set_dtype(model, 'bf16')
    try:
        b, _ = model._forward_encoder(feats.cuda(), lens)
        b = b.clone()
        twin = model.clone()
        assert twin.compute_dtype == 'bf16'
        c, _ = twin._forward_encoder(feats.cuda(), lens)
        assert torch.equal(b, c)
    finally:
        _set_dtype(model, 'fp32')
    d, _ = model._forward_encoder(feats.cuda(), lens)
    assert torch.equal(a, d)
    assert not torch.equal(a, b)
    assert (a - b).abs().max().item() < 0.2 * max(1.0, a.abs().max().item())
    with pytest.raises(ValueError):
        model.set_compute_dtype('fp16')


def test_bf16_whisper_golden_stays_close_to_the_fp32_reference():
    """Whisper-tiny-like encoder in bf16 against the REAL reference's committed fp32
    output: reported distance, bounded loosely (bf16 operands: ~3 significant
    digits per product)."""
    from golden_util import build_inputs, load_case
    meta, arrays = load_case('whisperenc_tiny')
    configs, sd, model = cached_model(meta['config'], meta['wseed'])
    _, _, feats, lens = build_inputs(meta)
    _set_dtype(model, 'bf16')
    try:
        enc, mask = model._forward_encoder(feats.cuda(), lens)
    finally:
        _set_dtype(model, 'fp32')
    enc = enc.cpu().numpy()
    for b, n in enumerate(arrays['enc_lens']):
        ref = arrays['enc_out'][b, :n]
        err = np.abs(enc[b, :n] - ref).max()
        assert err < 6e-2 * max(1.0, np.abs(ref).max()), (b, err)


@pytest.mark.parametrize('config,dma', [('whisper_tiny_like', 1), ('whisper_tiny_like', 0),
                                        ('tiny_sym', 0)])
def test_bf16_attention_large_score_range(config, dma):
    """Scores spanning hundreds of log2 units (query projection x 16): the online softmax only
    survives this if its running maximum is the TRUE maximum of the score tile -- a maximum
    taken from a partially written MFMA accumulator (the hazard the explicit v_max3 asm had no
    wait states for, round-4 advice) lets exp2 overflow: inf / NaN or a grossly wrong row.
    All three bf16 attention kernels (DMA-staged, register-staged, rel-pos)."""
    from wenet_amd import _lib, synthetic as S
    O = _oracle()
    configs = S.make_configs(config)
    sd = dict(S.make_state_dict(configs, 3))
    for k in list(sd):
        if k.endswith('self_attn.linear_q.weight') or k.endswith('self_attn.linear_q.bias'):
            sd[k] = sd[k] * 16.0
    from gpu_util import make_model
    model = make_model(configs, sd)
    feats, lens = S.make_features(3, (500, 900), seed=31, feat_dim=configs['input_dim'])
    with torch.no_grad(), O.bf16_operands(sd):
        ref, mask = O.encoder_forward(configs, sd, feats, lens, -1, -1)
    ref_lens = mask.squeeze(1).sum(1).numpy()
    L = _lib.lib()
    _set_dtype(model, 'bf16')
    try:
        _lib.check(L.wn_tune_set(b'attn_bf16_dma', dma), 'tune')
        enc, _ = model._forward_encoder(feats.cuda(), lens)
    finally:
        L.wn_tune_set(b'attn_bf16_dma', 1)
        _set_dtype(model, 'fp32')
    enc = enc.cpu()
    assert torch.isfinite(enc).all()
    for b in range(3):
        nb = int(ref_lens[b])
        scale = max(ref[b, :nb].abs().max().item(), 1.0)
        err = (enc[b, :nb] - ref[b, :nb]).abs().max().item() / scale
        print(f'{config} dma={dma} utt {b}: max rel err {err:.3e}')
        # peaked softmax rows: a probability on a bf16 rounding boundary moves a whole value
        # vector's weight by one bf16 ulp per layer
        assert err < 1.2e-2, (config, dma, b, err)   # measured 1.1e-3 (abs-pos) / 4.2e-3 (rel-pos)
