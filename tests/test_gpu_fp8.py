"""MXFP8 (OCP e4m3 + E8M0 block scales) path of the WN_PREC_FP8 mode -- BASELINE.json
configs[4] "MFMA fp8 FFN" -- through the C ABI against the oracle's rounding mode
(oracle/wenet_oracle.py: mx_quantize / bf16_operands(fp8_ffn=True)).

Bit-exact: the quantiser (element bytes and scale bytes) and the MXFP8 C of the GEMM
epilogue against the oracle's quantisation of the SAME kernel's fp32 C.  Tolerance:
the MFMA-scaled contraction against the fp64 product of the dequantised operands,
2e-5 x sum |a||w| (the block-scaled MFMA aligns the 64 products of a k-step to
their largest exponent before adding: measured 4.5e-6 with block scales spread over
2^+-6, r02e)."""
import numpy as np
import pytest
import torch

from gpu_util import cached_model

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import wenet_oracle as O
    return O


def _ptr(t):
    return None if t is None else t.data_ptr()


def _mxq_gpu(x):
    from wenet_amd import _lib
    L = _lib.lib()
    r, k = x.shape
    q = torch.empty((r, k), dtype=torch.uint8, device='cuda')
    sc = torch.zeros((k // 128, r), dtype=torch.int32, device='cuda')
    _lib.check(L.wn_op_mx_quantize(x.data_ptr(), r, k, q.data_ptr(), sc.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream), 'mxq')
    torch.cuda.synchronize()
    return q, sc


def _scale_bytes(sc, nblk):
    """dwords [K/128][rows] -> E bytes [rows][K/32]"""
    s = sc.cpu().numpy().astype(np.uint32)          # [kt][rows]
    out = np.zeros((s.shape[1], nblk), dtype=np.uint8)
    for kb in range(nblk):
        out[:, kb] = (s[kb // 4] >> (8 * (kb % 4))) & 0xff
    return out


def _gemm_mx(Aq, As, Wq, Ws, bias=None, resid=None, alpha=1.0, act=0, c_mode=0):
    from wenet_amd import _lib
    L = _lib.lib()
    M, K = Aq.shape
    N = Wq.shape[0]
    C = torch.empty((M, N), dtype=torch.uint8 if c_mode == 2 else torch.float32,
                    device='cuda')
    csc = torch.zeros(((N + 127) // 128, M), dtype=torch.int32, device='cuda')
    _lib.check(L.wn_op_gemm_lowp(_ptr(Aq), _ptr(Wq), _ptr(As), _ptr(Ws), _ptr(bias),
                                 _ptr(resid), _ptr(C), _ptr(csc), M, N, K, alpha, act,
                                 c_mode, 2, torch.cuda.current_stream().cuda_stream),
               'gemm_mxfp8')
    torch.cuda.synchronize()
    return C, csc


def test_mx_quantize_matches_oracle_bit_for_bit():
    O = _oracle()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(300, 512, generator=g) * torch.exp(torch.randn(300, 1, generator=g) * 3)
    x[0] = 0.0                                   # all-zero blocks
    x[1, :32] = 0.0
    x[2, 0] = 448.0 * 2.0 ** -5                  # amax exactly on a block-scale boundary
    x[2, 1:32] = 1e-4
    x[3, 0] = 449.0 * 2.0 ** 3                   # just above it
    x[4, :64] = torch.randn(64, generator=g) * 1e-30   # tiny values
    x[5, :32] = torch.linspace(-1, 1, 32) * 2.0 ** -9  # e4m3 subnormals after scaling
    x[5, 0] = 1.0
    q, sc = _mxq_gpu(x.cuda())
    rq, rE = O.mx_quantize(x)
    np.testing.assert_array_equal(_scale_bytes(sc, 16), rE.numpy())
    np.testing.assert_array_equal(q.cpu().numpy(), rq.view(torch.uint8).numpy())


@pytest.mark.parametrize('M,N,K,act,mode', [
    (512, 512, 256, 0, 'plain'),        # 2 K tiles
    (700, 520, 384, 1, 'plain'),        # ragged M / N, odd K-tile count
    (3000, 1280, 5120, 0, 'resid'),     # Whisper w_2
    (2900, 5120, 1280, 3, 'mx'),        # Whisper w_1: GELU, MXFP8 hidden
    (300, 288, 640, 2, 'mx'),
])
def test_gemm_mxfp8(M, N, K, act, mode):
    O = _oracle()
    from wenet_amd import _lib
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K + act)
    A = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))
    W = torch.randn(N, K, generator=g) * 0.2
    # make the block scales vary along k (catches a wrong op_sel / scale byte)
    A = A * torch.exp2(torch.randint(-6, 6, (1, K // 32), generator=g).float()
                       ).repeat_interleave(32, dim=1)
    bias = torch.randn(N, generator=g)
    resid = torch.randn(M, N, generator=g) if mode == 'resid' else None
    Aq, As = _mxq_gpu(A.cuda())
    Wq, Ws = _mxq_gpu(W.cuda())
    Ad = O.mx_round(A).double()
    Wd = O.mx_round(W).double()
    y = (Ad @ Wd.T).float() + bias
    if act == 1:
        y = torch.nn.functional.silu(y)
    elif act == 2:
        y = torch.relu(y)
    elif act == 3:
        y = torch.nn.functional.gelu(y)
    y = 0.5 * y
    if resid is not None:
        y = y + resid
    scale = (Ad.abs() @ Wd.abs().T).max().item()
    args = (Aq, As, Wq, Ws, bias.cuda(), resid.cuda() if resid is not None else None,
            0.5, act)
    got, _ = _gemm_mx(*args, c_mode=0)
    for _ in range(3):                                 # race screen
        again, _ = _gemm_mx(*args, c_mode=0)
        assert torch.equal(got, again), 'MXFP8 GEMM is not deterministic (race?)'
    if mode == 'mx':
        # fp32 C of the same accumulators: without residual
        err = (got.cpu() - y).abs().max().item()
        assert err <= 2e-5 * scale + 2e-5, (err, scale)
        cq, csc = _gemm_mx(*args, c_mode=2)
        rq, rE = O.mx_quantize(got.cpu())
        np.testing.assert_array_equal(_scale_bytes(csc, N // 32), rE.numpy())
        np.testing.assert_array_equal(cq.cpu().numpy(), rq.view(torch.uint8).numpy())
    else:
        err = (got.cpu() - y).abs().max().item()
        assert err <= 2e-5 * scale + 2e-5, (err, scale)


@pytest.mark.parametrize('B,frames', [(3, (1400, 1500))])
def test_fp8_whisper_encoder_vs_oracle_rounding_mode(B, frames):
    """WN_PREC_FP8 through the model: the Whisper-large-width encoder (2 blocks) with
    MXFP8 feed-forward GEMMs against the oracle under bf16_operands(fp8_ffn=True).
    The product quantises the SAME fp32 tensors the oracle does only up to the bf16 /
    e4m3 rounding of upstream results, so single elements may land on the other side
    of a rounding boundary: tolerance on the encoder output 3e-2 of its scale (bf16
    mode: 6e-3 .. 2e-2), and the fp8 mode must differ from the bf16 mode (it ran)."""
    from wenet_amd import synthetic as S
    O = _oracle()
    configs, sd, model = cached_model('whisper_largev3_2blocks', 0)
    feats, lens = S.make_features(B, frames, seed=31, feat_dim=configs['input_dim'])
    from wenet_amd import _lib
    L = _lib.lib()
    try:
        _lib.check(L.wn_tune_set(b'fp8_min_tiles', 0), 'tune')   # small test batch
        model.set_compute_dtype('fp8')
        assert model.compute_dtype == 'fp8'
        enc8, mask = model._forward_encoder(feats.cuda(), lens)
        enc8 = enc8.cpu()
        model.set_compute_dtype('bf16')
        enc16, _ = model._forward_encoder(feats.cuda(), lens)
        enc16 = enc16.cpu()
    finally:
        model.set_compute_dtype('fp32')
        L.wn_tune_set(b'fp8_min_tiles', 192)
    with torch.no_grad(), O.bf16_operands(sd, fp8_ffn=True):
        ref, rmask = O.encoder_forward(configs, sd, feats, lens)
    n = rmask.squeeze(1).sum(1)
    scale = ref.abs().max().item()
    worst = 0.0
    for b in range(B):
        worst = max(worst, (enc8[b, :n[b]] - ref[b, :n[b]]).abs().max().item())
    print(f'\nfp8 encoder vs oracle(fp8_ffn): max err {worst:.3e} of scale {scale:.3f}; '
          f'fp8 vs bf16 mode: {(enc8 - enc16).abs().max().item():.3e}')
    assert worst < 3e-2 * scale
    assert (enc8 - enc16).abs().max().item() > 1e-4 * scale


@pytest.mark.parametrize('config,B,frames', [('aishell_u2pp', 4, (500, 700))])
def test_fp8_conformer_encoder_every_ffn_takes_the_mx_path(config, B, frames):
    """WN_PREC_FP8 on a CONFORMER: both feed-forward modules of every layer (macaron and
    final) run their two GEMMs on MXFP8 operands -- in the bf16 / fp32 modes the macaron
    module of layers 1.. gets LN(x) from the previous layer's fused tail, in the fp8 mode every
    module normalises for itself (layernorm_mx) so that none silently stays on bf16.  Against
    the oracle under bf16_operands(fp8_ffn=True), which rounds EVERY encoder w_1 / w_2 operand
    pair to MXFP8; same tolerance as the Whisper-width test; and the fp8 output differs from the
    bf16 mode's by more than one module's worth of e4m3 noise."""
    from wenet_amd import synthetic as S
    O = _oracle()
    configs, sd, model = cached_model(config, 0)
    feats, lens = S.make_features(B, frames, seed=77)
    from wenet_amd import _lib
    L = _lib.lib()
    try:
        _lib.check(L.wn_tune_set(b'fp8_min_tiles', 0), 'tune')   # small test batch
        model.set_compute_dtype('fp8')
        enc8, mask = model._forward_encoder(feats.cuda(), lens)
        enc8 = enc8.cpu()
        model.set_compute_dtype('bf16')
        enc16, _ = model._forward_encoder(feats.cuda(), lens)
        enc16 = enc16.cpu()
    finally:
        model.set_compute_dtype('fp32')
        L.wn_tune_set(b'fp8_min_tiles', 192)
    with torch.no_grad(), O.bf16_operands(sd, fp8_ffn=True):
        ref8, rmask = O.encoder_forward(configs, sd, feats, lens)
    with torch.no_grad(), O.bf16_operands(sd):
        ref16, _ = O.encoder_forward(configs, sd, feats, lens)
    n = rmask.squeeze(1).sum(1)
    scale = ref8.abs().max().item()
    e8 = max((enc8[b, :n[b]] - ref8[b, :n[b]]).abs().max().item() for b in range(B))
    e16 = max((enc16[b, :n[b]] - ref16[b, :n[b]]).abs().max().item() for b in range(B))
    d_ref = max((ref8[b, :n[b]] - ref16[b, :n[b]]).abs().max().item() for b in range(B))
    d_gpu = max((enc8[b, :n[b]] - enc16[b, :n[b]]).abs().max().item() for b in range(B))
    print(f'\n[{config}] fp8 vs oracle(fp8_ffn) {e8:.3e}, bf16 vs oracle(bf16) {e16:.3e} of scale '
          f'{scale:.3f}; fp8 - bf16: oracle {d_ref:.3e}, GPU {d_gpu:.3e}')
    assert e8 < 3e-2 * scale and e16 < 2e-2 * scale
    # all 24 modules quantised: the GPU's fp8 - bf16 distance is of the oracle's size (a path
    # that quantised only some modules would sit well below it)
    assert d_gpu > 0.5 * d_ref
