// fp32 GEMM on CDNA4 matrix cores: C = epi(A[M,K] * W[N,K]^T).
//
// This is the kernel family that carries ~98 % of the encoder's FLOPs (FFN,
// QKV / out projections, pointwise convs, CTC head, decoder projections and the
// implicit-GEMM 3x3/s2 subsampling conv).  It replaces the torch.nn.Linear /
// Conv calls of wenet/models/transformer/{positionwise_feed_forward.py:50-58,
// attention.py:109-131,176, convolution.py:138,148, subsampling.py:188-226}.
//
// Design (gfx950):
//  * v_mfma_f32_32x32x2_f32: exact fp32 multiply / fp32 accumulate (bitwise an
//    fmaf chain), 64 FLOP/clk/SIMD = 157.3 TF chip peak.  The reference runs
//    fp32, and greedy-token identity / 1e-3 rescoring parity need fp32.
//  * 256 threads = 4 waves in a WGM x WGN grid; each wave owns MT x NT 32x32
//    accumulator tiles (64 acc VGPRs for the 128x128 block).
//  * K tile = 32 floats.  A and W tiles are staged global -> VGPR -> LDS with a
//    36-float row stride: a wave's ds_read_b128 (lane = row, 16 B of k) then
//    hits 16 distinct 16-B slots per 16-lane service group -> conflict-free.
//    One 16-B LDS read feeds 4 MFMAs (lanes 0-31 carry k..k+3, lanes 32-63
//    k+4..k+7; any k pairing is legal as long as A and W agree).
//  * double-buffered LDS, one barrier per K tile; global loads run two tiles
//    ahead (two named register sets) so that even a 16-MFMA tile hides an
//    Infinity-Cache / HBM round trip.
//  * XCD-aware block order: the N tiles of one M panel run on the same XCD so
//    the A panel is fetched into one L2, not eight.
//  * epilogue fused: bias, SiLU/ReLU, alpha, residual add, GLU.
#include "common.h"
#include "gemm_epilogue.h"

namespace wn {

thread_local int t_gemm_prec = PREC_F32;

namespace {

constexpr int BK = 32;  // K granularity every problem must respect

template <int BM, int BN, int WGM, int WGN, int ACT, bool RESID, bool GLU,
          bool CONV, int BK = 32, int PF = 1>
__global__ __launch_bounds__(WGM * WGN * 64) void gemm_f32_kernel(
    GemmArgs p, int tiles_m, int tiles_n) {
  constexpr int LDS_STRIDE = BK + 4;  // floats; rows land on distinct 16-B slots
  constexpr int KC = BK / 4;          // float4 chunks per tile row
  constexpr int NTHR = WGM * WGN * 64;
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int A_CHUNKS = BM * KC / NTHR;  // float4 chunks per thread
  constexpr int B_CHUNKS = BN * KC / NTHR;
  static_assert(A_CHUNKS >= 1 && B_CHUNKS >= 1 && MT >= 1 && NT >= 1, "tile");
  static_assert(!GLU || NT == 2, "GLU epilogue needs a 64-wide wave tile");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TILE = (BM + BN) * LDS_STRIDE;  // floats per buffer: A then W

  // ---- XCD-aware tile assignment (gemm_epilogue.h) -------------------------
  const int bid = xcd_block_order(blockIdx.x, tiles_m * tiles_n);
  const int tm = bid / tiles_n, tn = bid % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN, wn_ = wave % WGN;
  // ---- per-thread global load descriptors --------------------------------
  const float* a_ptr[A_CHUNKS];
  int a_lds[A_CHUNKS];
#pragma unroll
  for (int i = 0; i < A_CHUNKS; ++i) {
    const int c = tid + NTHR * i;
    const int row = c / KC, kc = c % KC;
    int grow = m0 + row;
    if (grow > p.M - 1) grow = p.M - 1;
    if (CONV) {
      a_ptr[i] = p.A + p.a_row_off[grow] + kc * 4;
    } else {
      a_ptr[i] = p.A + (int64_t)grow * p.lda + kc * 4;
    }
    a_lds[i] = row * LDS_STRIDE + kc * 4;
  }
  const float* b_ptr[B_CHUNKS];
  int b_lds[B_CHUNKS];
#pragma unroll
  for (int i = 0; i < B_CHUNKS; ++i) {
    const int c = tid + NTHR * i;
    const int row = c / KC, kc = c % KC;
    int grow = n0 + row;
    if (grow > p.N - 1) grow = p.N - 1;
    b_ptr[i] = p.W + (int64_t)grow * p.K + kc * 4;
    b_lds[i] = row * LDS_STRIDE + kc * 4;
  }

  auto gload = [&](int kt, f32x4 (&ra)[A_CHUNKS], f32x4 (&rb)[B_CHUNKS]) {
    const int k0 = kt * BK;
    int64_t aoff = k0;
    if (CONV) {
      const int tap = k0 / p.conv_C;
      aoff = (int64_t)(tap / 3) * p.conv_sy + (int64_t)(tap % 3) * p.conv_sx +
             (k0 - tap * p.conv_C);
    }
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i)
      ra[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + aoff);
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i)
      rb[i] = *reinterpret_cast<const f32x4*>(b_ptr[i] + k0);
  };
  auto lstore = [&](int buf, const f32x4 (&ra)[A_CHUNKS],
                    const f32x4 (&rb)[B_CHUNKS]) {
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i)
      *reinterpret_cast<f32x4*>(smem + buf * TILE + a_lds[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i)
      *reinterpret_cast<f32x4*>(smem + buf * TILE + BM * LDS_STRIDE +
                                b_lds[i]) = rb[i];
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // LDS fragment addresses: row = lane & 31, k sub-block = lane >> 5.
  const int frag_off = (lane & 31) * LDS_STRIDE + (lane >> 5) * 4;
  const int a_frag = (wm * WTM) * LDS_STRIDE + frag_off;
  const int b_frag = (wn_ * WTN) * LDS_STRIDE + frag_off;
  auto compute = [&](int cur) {
    const float* cA = smem + cur * TILE + a_frag;
    const float* cB = smem + cur * TILE + BM * LDS_STRIDE + b_frag;
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      f32x4 fa[MT], fb[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i)
        fa[i] = *reinterpret_cast<const f32x4*>(cA + i * 32 * LDS_STRIDE +
                                                kk * 8);
#pragma unroll
      for (int j = 0; j < NT; ++j)
        fb[j] = *reinterpret_cast<const f32x4*>(cB + j * 32 * LDS_STRIDE +
                                                kk * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
    }
  };

  const int nk = p.K / BK;
  if constexpr (PF == 2) {
    // Global prefetch distance 2 (two register sets, statically named): the
    // loads of tile kt+2 are issued before the MFMAs of tile kt and consumed
    // one whole iteration later.
    f32x4 ra0[A_CHUNKS], rb0[B_CHUNKS], ra1[A_CHUNKS], rb1[B_CHUNKS];
    gload(0, ra0, rb0);
    if (nk > 1) gload(1, ra1, rb1);
    lstore(0, ra0, rb0);
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      if (kt + 2 < nk) gload(kt + 2, ra0, rb0);
      compute(0);
      lstore(1, ra1, rb1);          // tile kt+1, loaded one iteration ago
      __syncthreads();
      if (kt + 3 < nk) gload(kt + 3, ra1, rb1);
      compute(1);
      if (kt + 2 < nk) lstore(0, ra0, rb0);
      __syncthreads();
    }
    if (kt < nk) compute(0);        // odd tile count: the last tile is in buf 0
  } else {
    f32x4 ra[A_CHUNKS], rb[B_CHUNKS];
    gload(0, ra, rb);
    lstore(0, ra, rb);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) gload(kt + 1, ra, rb);
      compute(cur);
      if (kt + 1 < nk) lstore(cur ^ 1, ra, rb);
      __syncthreads();
    }
  }

  gemm_epilogue<BM, BN, WGM, WGN, ACT, RESID, GLU>(p, acc, m0, n0, wm, wn_, lane);
}

template <int BM, int BN, int WGM, int WGN, int ACT, bool RESID, bool GLU,
          bool CONV, int BKT = 32, int PF = 1>
int launch(const GemmArgs& a, hipStream_t stream) {
  const int tiles_m = cdiv(a.M, BM), tiles_n = cdiv(a.N, BN);
  const size_t lds = 2 * (BM + BN) * (BKT + 4) * sizeof(float);
  auto kern = gemm_f32_kernel<BM, BN, WGM, WGN, ACT, RESID, GLU, CONV, BKT, PF>;
  WN_MAX_DYN_LDS(kern, lds);
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(WGM * WGN * 64), lds,
                     stream, a, tiles_m, tiles_n);
  WN_HIP(hipGetLastError());
  return 0;
}

// GLU_ONLY: the block shapes with a 64-wide wave tile exist for the GLU epilogue alone; every
// other shape is instantiated for the plain epilogues only (the kernels the tile rule below can
// reach, nothing else)
template <int BM, int BN, int WGM, int WGN, bool CONV, bool GLU_ONLY = false>
int dispatch_epi(const GemmArgs& a, hipStream_t s) {
  const bool resid = a.resid != nullptr;
  // long K loops stream an operand from HBM / Infinity Cache: prefetch two
  // tiles ahead there (measured +2..4 %); short ones (K = d) gain nothing
  const bool pf2 = !CONV && a.K >= 1024;
  if constexpr (GLU_ONLY) {
    static_assert(BN / WGN == 64 && !CONV, "GLU epilogue needs a 64-wide wave tile");
    if (a.glu) return launch<BM, BN, WGM, WGN, ACT_NONE, false, true, false>(a, s);
    set_error("gemm: this block shape is built for the GLU epilogue only");
    return -1;
  } else {
  if (a.glu) {
    set_error("gemm: GLU epilogue needs a 64-wide wave tile");
    return -1;
  }
  switch (a.act) {
    case ACT_NONE:
      if (pf2)
        return resid ? launch<BM, BN, WGM, WGN, ACT_NONE, true, false, CONV, 32, 2>(a, s)
                     : launch<BM, BN, WGM, WGN, ACT_NONE, false, false, CONV, 32, 2>(a, s);
      return resid ? launch<BM, BN, WGM, WGN, ACT_NONE, true, false, CONV>(a, s)
                   : launch<BM, BN, WGM, WGN, ACT_NONE, false, false, CONV>(a, s);
    case ACT_SILU:
      if constexpr (!CONV)
        return resid ? launch<BM, BN, WGM, WGN, ACT_SILU, true, false, false>(a, s)
                     : launch<BM, BN, WGM, WGN, ACT_SILU, false, false, false>(a, s);
      break;
    case ACT_GELU:
      return resid ? launch<BM, BN, WGM, WGN, ACT_GELU, true, false, CONV>(a, s)
                   : launch<BM, BN, WGM, WGN, ACT_GELU, false, false, CONV>(a, s);
    case ACT_RELU:
      if (pf2)
        return resid ? launch<BM, BN, WGM, WGN, ACT_RELU, true, false, CONV, 32, 2>(a, s)
                     : launch<BM, BN, WGM, WGN, ACT_RELU, false, false, CONV, 32, 2>(a, s);
      return resid ? launch<BM, BN, WGM, WGN, ACT_RELU, true, false, CONV>(a, s)
                   : launch<BM, BN, WGM, WGN, ACT_RELU, false, false, CONV>(a, s);
  }
  }
  set_error("gemm: unsupported epilogue");
  return -1;
}

}  // namespace

int gemm_f32(const GemmArgs& a, hipStream_t stream) {
  WN_CHECK(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem");
  WN_CHECK(a.K % BK == 0, "gemm: K must be a multiple of 32");
  WN_CHECK(a.A && a.W && a.C, "gemm: null operand");
  const bool conv = a.a_row_off != nullptr;
  if (conv) {
    WN_CHECK(a.conv_C % BK == 0 && (a.K == 9 * a.conv_C || a.K == a.conv_C),
             "gemm(conv): K must be 9*C (3x3 taps) or C (gathered rows), C % 32 == 0");
  } else {
    WN_CHECK(a.lda % 4 == 0, "gemm: lda must be a multiple of 4 floats");
  }
  if (a.glu) WN_CHECK(a.N % 64 == 0, "gemm(GLU): N must be a multiple of 64");
  if (t_gemm_prec == PREC_BF16) return gemm_bf16(a, stream);
  WN_CHECK(!a.a_bf16 && !a.c_bf16, "gemm: bf16 operands need the bf16 mode");
  // Tile choice (measured on M = 7932 rows, profiles/): 8 waves per block hide
  // the barrier / LDS latency of the short K loops better than 4; the block
  // shrinks with the problem so that the grid still covers the 256 CUs.
  //   1: 128x128, 2x4 waves   2: 128x128, 4x2 waves (64-wide wave tile: GLU)
  //   3: 64x128, 2x4 waves    4: 64x128, 2x2 waves (GLU)   5: 64x64, 2x2 waves
  //   6: 128x128, 2x2 waves (conv)
  // (a 256x256 one-block-per-CU tile was +3.5 % on the isolated FFN-w1 shape and -14 % inside
  // the decode pipeline, where the other stream's search kernel holds CUs: removed, docs/LOG_rounds1-3.md section 6)
  const int64_t t128 = (int64_t)cdiv(a.M, 128) * cdiv(a.N, 128);
  const int64_t t64x128 = (int64_t)cdiv(a.M, 64) * cdiv(a.N, 128);
  if (a.glu)
    return t128 >= 224 ? dispatch_epi<128, 128, 4, 2, false, true>(a, stream)
                       : dispatch_epi<64, 128, 2, 2, false, true>(a, stream);
  if (conv)   // K = 9C: the 4-wave block wins
    return t128 >= 384 ? dispatch_epi<128, 128, 2, 2, true>(a, stream)
                       : dispatch_epi<64, 128, 2, 2, true>(a, stream);
  if (t128 >= 384) return dispatch_epi<128, 128, 2, 4, false>(a, stream);
  if (t64x128 >= 384) return dispatch_epi<64, 128, 2, 4, false>(a, stream);
  return dispatch_epi<64, 64, 2, 2, false>(a, stream);
}

}  // namespace wn
