// bf16-operand GEMM on CDNA4 matrix cores: C = epi(bf16(A[M,K]) * bf16(W[N,K])^T),
// fp32 accumulate, fp32 C.
//
// This is the opt-in reduced-precision mode of the decode path -- the reference's
// `recognize.py --dtype bf16` (wenet/bin/recognize.py:52-56,250-255,278-280:
// torch autocast around model.decode) and BASELINE.json configs[4] (Whisper
// large-v3 encoder in bf16).  Autocast rounds the operands AND the result of every
// linear / conv to bf16; here only the operands are rounded (RNE,
// v_cvt_pk_bf16_f32), the accumulation and every activation tensor stay fp32, so
// this mode is at least as precise as the reference's.  The default path
// (gemm.hip, fp32 operands) is untouched.
//
// Design (gfx950):
//  * v_mfma_f32_32x32x16_bf16: 16x the rate of the fp32-operand MFMA (2.5 PF
//    dense chip peak).  A lane holds 8 consecutive k of row (lane & 31), lanes
//    32..63 the next 8; A and W use the same k assignment, so the contraction
//    is independent of the hardware's k order.
//  * activations and weights stay fp32 in HBM (same buffers as the fp32 mode):
//    the fp32 -> bf16 conversion happens in registers between the global load
//    and the LDS store, so the LDS tile and the LDS -> register traffic are
//    half the fp32 kernel's and one ds_read_b128 feeds one 32-cycle MFMA.
//  * K tile of 64 (32 when K or the conv channel count is not a multiple of 64),
//    row stride padded by 16 B: a wave's ds_read_b128 (lane = row) lands on
//    distinct 16-B slots.
//  * same block shapes, XCD-aware block order and fused epilogue as gemm.hip
//    (gemm_epilogue.h).
// With fp32 operands in HBM / L2 the kernel is bound by the global -> LDS stream
// (8 B per bf16 MAC pair), not by the matrix cores; bf16 activation storage is
// the next step (docs/LOG_rounds1-3.md section 7).
#include "common.h"
#include "gemm_epilogue.h"

namespace wn {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x4 to_bf16(const f32x4& v) {
  bf16x4 r;
  r[0] = (__bf16)v[0]; r[1] = (__bf16)v[1];  // v_cvt_pk_bf16_f32: round to
  r[2] = (__bf16)v[2]; r[3] = (__bf16)v[3];  // nearest even
  return r;
}

template <int BM, int BN, int WGM, int WGN, int ACT, bool RESID, bool GLU,
          bool CONV, int BK>
__global__ __launch_bounds__(WGM * WGN * 64) void gemm_bf16_kernel(
    GemmArgs p, int tiles_m, int tiles_n) {
  constexpr int LDS_STRIDE = BK + 8;  // bf16 elements; +16 B per row
  constexpr int KC = BK / 4;          // 4-element chunks per tile row
  constexpr int NTHR = WGM * WGN * 64;
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int A_CHUNKS = BM * KC / NTHR;  // float4 loads per thread
  constexpr int B_CHUNKS = BN * KC / NTHR;
  static_assert(A_CHUNKS >= 1 && B_CHUNKS >= 1 && MT >= 1 && NT >= 1, "tile");
  static_assert(!GLU || NT == 2, "GLU epilogue needs a 64-wide wave tile");
  static_assert(BK % 16 == 0, "K tile is a multiple of the MFMA K");
  extern __shared__ __attribute__((aligned(16))) __bf16 smem_h[];
  constexpr int TILE = (BM + BN) * LDS_STRIDE;  // elements per buffer: A then W

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

  // global fp32 -> registers; the conversion waits until the LDS store, after
  // the MFMAs of the current tile, so the loads stay in flight under them
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
      *reinterpret_cast<bf16x4*>(smem_h + buf * TILE + a_lds[i]) = to_bf16(ra[i]);
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i)
      *reinterpret_cast<bf16x4*>(smem_h + buf * TILE + BM * LDS_STRIDE +
                                 b_lds[i]) = to_bf16(rb[i]);
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // LDS fragment addresses: row = lane & 31, 8 consecutive k at (lane >> 5) * 8.
  const int frag_off = (lane & 31) * LDS_STRIDE + (lane >> 5) * 8;
  const int a_frag = (wm * WTM) * LDS_STRIDE + frag_off;
  const int b_frag = (wn_ * WTN) * LDS_STRIDE + frag_off;
  auto compute = [&](int cur) {
    const __bf16* cA = smem_h + cur * TILE + a_frag;
    const __bf16* cB = smem_h + cur * TILE + BM * LDS_STRIDE + b_frag;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8 fa[MT], fb[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i)
        fa[i] = *reinterpret_cast<const bf16x8*>(cA + i * 32 * LDS_STRIDE +
                                                 kk * 16);
#pragma unroll
      for (int j = 0; j < NT; ++j)
        fb[j] = *reinterpret_cast<const bf16x8*>(cB + j * 32 * LDS_STRIDE +
                                                 kk * 16);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  };

  const int nk = p.K / BK;
  {
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
          bool CONV, int BKT>
int launch(const GemmArgs& a, hipStream_t stream) {
  const int tiles_m = cdiv(a.M, BM), tiles_n = cdiv(a.N, BN);
  const size_t lds = 2 * (BM + BN) * (BKT + 8) * sizeof(__bf16);
  auto kern = gemm_bf16_kernel<BM, BN, WGM, WGN, ACT, RESID, GLU, CONV, BKT>;
  WN_MAX_DYN_LDS(kern, lds);
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(WGM * WGN * 64), lds,
                     stream, a, tiles_m, tiles_n);
  WN_HIP(hipGetLastError());
  return 0;
}

// GLU_ONLY: as in gemm.hip -- only the kernels the tile rule below can reach are instantiated
template <int BM, int BN, int WGM, int WGN, bool CONV, int BKT, bool GLU_ONLY = false>
int dispatch_epi(const GemmArgs& a, hipStream_t s) {
  const bool resid = a.resid != nullptr;
  if constexpr (GLU_ONLY) {
    static_assert(BN / WGN == 64 && !CONV, "GLU epilogue needs a 64-wide wave tile");
    if (a.glu) return launch<BM, BN, WGM, WGN, ACT_NONE, false, true, false, BKT>(a, s);
    set_error("gemm(bf16): this block shape is built for the GLU epilogue only");
    return -1;
  } else {
  if (a.glu) {
    set_error("gemm(bf16): GLU epilogue needs a 64-wide wave tile");
    return -1;
  }
  switch (a.act) {
    case ACT_NONE:
      return resid ? launch<BM, BN, WGM, WGN, ACT_NONE, true, false, CONV, BKT>(a, s)
                   : launch<BM, BN, WGM, WGN, ACT_NONE, false, false, CONV, BKT>(a, s);
    case ACT_SILU:
      if constexpr (!CONV)
        return resid ? launch<BM, BN, WGM, WGN, ACT_SILU, true, false, false, BKT>(a, s)
                     : launch<BM, BN, WGM, WGN, ACT_SILU, false, false, false, BKT>(a, s);
      break;
    case ACT_GELU:
      return resid ? launch<BM, BN, WGM, WGN, ACT_GELU, true, false, CONV, BKT>(a, s)
                   : launch<BM, BN, WGM, WGN, ACT_GELU, false, false, CONV, BKT>(a, s);
    case ACT_RELU:
      return resid ? launch<BM, BN, WGM, WGN, ACT_RELU, true, false, CONV, BKT>(a, s)
                   : launch<BM, BN, WGM, WGN, ACT_RELU, false, false, CONV, BKT>(a, s);
  }
  }
  set_error("gemm(bf16): unsupported epilogue");
  return -1;
}

template <int BKT>
int dispatch_tile(const GemmArgs& a, bool conv, hipStream_t stream) {
  // Block shapes as in gemm.hip; with 16x the MFMA rate the global -> LDS
  // stream bounds the kernel, so the largest block that still fills the chip
  // is preferred (bytes per MAC fall with the tile edge).
  //   1: 128x128, 2x4 waves   2: 128x128, 4x2 waves (64-wide wave tile: GLU)
  //   4: 64x128, 2x2 waves (GLU, conv)   5: 64x64, 2x2 waves
  //   7: 256x256, 4x2 waves (144 KB LDS, one block per CU, K tile 64 only)
  // Measured (tools/bench_gemm.py --bf16, profiles/r01g_gemm_bf16_tiles.txt):
  // 256x256 wins by 28-47 % once it fills the chip for more than ~2 rounds or
  // the K loop is long (M=24000: N=5120/3840 K=1280, N=1280 K=5120), loses 4 %
  // at 470 tiles with K=1280, and badly when it cannot cover the CUs.
  const int64_t t128 = (int64_t)cdiv(a.M, 128) * cdiv(a.N, 128);
  const int64_t t256 = (int64_t)cdiv(a.M, 256) * cdiv(a.N, 256);
  // gemm_tile_bf16 = 1 (tests: the pipelined kernel against these): never the 256-row tile
  if (a.glu)
    return t128 >= 224 ? dispatch_epi<128, 128, 4, 2, false, BKT, true>(a, stream)
                       : dispatch_epi<64, 128, 2, 2, false, BKT, true>(a, stream);
  if (conv)
    return t128 >= 224 ? dispatch_epi<128, 128, 2, 4, true, BKT>(a, stream)
                       : dispatch_epi<64, 128, 2, 2, true, BKT>(a, stream);
  if constexpr (BKT == 64) {
    if (tune().gemm_tile_bf16 != 1 && t256 >= 256 && (t256 >= 1024 || a.K >= 2048))
      return dispatch_epi<256, 256, 4, 2, false, 64>(a, stream);
  }
  return t128 >= 224 ? dispatch_epi<128, 128, 2, 4, false, BKT>(a, stream)
                     : dispatch_epi<64, 64, 2, 2, false, BKT>(a, stream);
}

}  // namespace

thread_local const float* t_wslab_f32 = nullptr;
thread_local const void* t_wslab_bf16 = nullptr;
thread_local int64_t t_wslab_elems = 0;

int gemm_bf16(const GemmArgs& a, hipStream_t stream) {
  // argument checks are gemm_f32's (the only caller)
  if (a.a_bf16) {
    WN_CHECK(t_wslab_bf16 && a.W >= t_wslab_f32 && a.W < t_wslab_f32 + t_wslab_elems,
             "gemm(bf16 stored): W is not inside the handle's converted weight slab");
    const char* wh = reinterpret_cast<const char*>(t_wslab_bf16) +
                     (a.W - t_wslab_f32) * 2;
    return gemm_bf16_stored(a, wh, stream);
  }
  WN_CHECK(!a.c_bf16, "gemm(bf16): bf16 C needs bf16 A");
  const bool conv = a.a_row_off != nullptr;
  const bool k64 = a.K % 64 == 0 && (!conv || a.conv_C % 64 == 0);
  return k64 ? dispatch_tile<64>(a, conv, stream) : dispatch_tile<32>(a, conv, stream);
}

}  // namespace wn
