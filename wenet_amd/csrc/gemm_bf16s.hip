// bf16-STORED operands: C = epi(A[M,K] * W[N,K]^T) with A and W already bf16 in HBM,
// fp32 accumulate, C fp32 or bf16.  The storage form of the WN_PREC_BF16 mode
// (gemm_bf16.hip converts fp32 operands on the fly): the tensors that only ever
// feed a GEMM -- LayerNorm output, FFN hidden, attention context -- are written as
// bf16 by their producers, and the weight slab has a one-time bf16 copy.  The
// arithmetic is identical to gemm_bf16.hip (its first step is the same rounding,
// and rounding is idempotent); what changes is the traffic and the pipeline:
//  * half the operand bytes from HBM / L2, no conversion instructions;
//  * a K tile is 16 + 16 staging VGPRs per thread (256x256 block) instead of 64,
//    so the global loads run TWO tiles ahead (two named register sets, as the
//    fp32 kernel's PF = 2 path) -- the exposed L2 / Infinity-Cache latency was what
//    bounded gemm_bf16_kernel (docs/LOG_rounds1-3.md section 3);
//  * 16-byte loads and 16-byte LDS stores on both operands.
// Same block shapes, LDS image (row stride K tile + 8 elements), fragment reads,
// XCD order and epilogue as gemm_bf16.hip.
#include "common.h"
#include "gemm_epilogue.h"

namespace wn {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int BM, int BN, int WGM, int WGN, int ACT, bool RESID, bool GLU, bool CH,
          int BK>
__global__ __launch_bounds__(WGM * WGN * 64) void gemm_bf16s_kernel(
    GemmArgs p, const __bf16* __restrict__ Wh, int tiles_m, int tiles_n) {
  constexpr int LDS_STRIDE = BK + 8;  // bf16 elements; +16 B per row
  constexpr int KC = BK / 8;          // 8-element (16-byte) chunks per tile row
  constexpr int NTHR = WGM * WGN * 64;
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int A_CHUNKS = BM * KC / NTHR;
  constexpr int B_CHUNKS = BN * KC / NTHR;
  static_assert(A_CHUNKS >= 1 && B_CHUNKS >= 1 && MT >= 1 && NT >= 1, "tile");
  static_assert(!GLU || NT == 2, "GLU epilogue needs a 64-wide wave tile");
  extern __shared__ __attribute__((aligned(16))) __bf16 smem_s[];
  constexpr int TILE = (BM + BN) * LDS_STRIDE;  // elements per buffer: A then W

  const int bid = xcd_block_order(blockIdx.x, tiles_m * tiles_n);
  const int tm = bid / tiles_n, tn = bid % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN, wn_ = wave % WGN;
  // Per-thread chunk addresses as 32-bit BYTE offsets from the two (uniform)
  // base pointers: 1 VGPR per chunk instead of a 64-bit pointer, and the LDS
  // offsets of chunk i are compile-time distances from chunk 0 -- the 256x256
  // block holds 128 accumulators + two staging sets and has no register to spare.
  static_assert(NTHR % KC == 0, "chunk rows advance uniformly");
  constexpr int RSTEP = NTHR / KC;  // rows between consecutive chunks of a thread
  const char* Ab = reinterpret_cast<const char*>(p.A);  // bf16 [M][lda]
  const char* Wb = reinterpret_cast<const char*>(Wh);   // bf16 [N][K]
  const int row0 = tid / KC, kc0 = tid % KC;
  const int lds0 = row0 * LDS_STRIDE + kc0 * 8;
  unsigned a_off[A_CHUNKS], b_off[B_CHUNKS];
#pragma unroll
  for (int i = 0; i < A_CHUNKS; ++i) {
    int grow = m0 + row0 + i * RSTEP;
    if (grow > p.M - 1) grow = p.M - 1;
    a_off[i] = ((unsigned)grow * (unsigned)p.lda + (unsigned)kc0 * 8u) * 2u;
  }
#pragma unroll
  for (int i = 0; i < B_CHUNKS; ++i) {
    int grow = n0 + row0 + i * RSTEP;
    if (grow > p.N - 1) grow = p.N - 1;
    b_off[i] = ((unsigned)grow * (unsigned)p.K + (unsigned)kc0 * 8u) * 2u;
  }

  auto gload = [&](int kt, bf16x8 (&ra)[A_CHUNKS], bf16x8 (&rb)[B_CHUNKS]) {
    const unsigned k0b = (unsigned)(kt * BK) * 2u;
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i)
      ra[i] = *reinterpret_cast<const bf16x8*>(Ab + (a_off[i] + k0b));
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i)
      rb[i] = *reinterpret_cast<const bf16x8*>(Wb + (b_off[i] + k0b));
  };
  auto lstore = [&](int buf, const bf16x8 (&ra)[A_CHUNKS],
                    const bf16x8 (&rb)[B_CHUNKS]) {
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i)
      *reinterpret_cast<bf16x8*>(smem_s + buf * TILE + lds0 + i * RSTEP * LDS_STRIDE) =
          ra[i];
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i)
      *reinterpret_cast<bf16x8*>(smem_s + buf * TILE + BM * LDS_STRIDE + lds0 +
                                 i * RSTEP * LDS_STRIDE) = rb[i];
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
    const __bf16* cA = smem_s + cur * TILE + a_frag;
    const __bf16* cB = smem_s + cur * TILE + BM * LDS_STRIDE + b_frag;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8 fa[MT], fb[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i)
        fa[i] = *reinterpret_cast<const bf16x8*>(cA + i * 32 * LDS_STRIDE + kk * 16);
#pragma unroll
      for (int j = 0; j < NT; ++j)
        fb[j] = *reinterpret_cast<const bf16x8*>(cB + j * 32 * LDS_STRIDE + kk * 16);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j],
                                                              0, 0, 0);
    }
  };

  // Global prefetch distance 2 (two register sets, statically named): the loads
  // of tile kt+2 are issued before the MFMAs of tile kt and consumed one whole
  // iteration later (the loop of gemm_f32_kernel's PF == 2 path).
  const int nk = p.K / BK;
  {
    bf16x8 ra0[A_CHUNKS], rb0[B_CHUNKS], ra1[A_CHUNKS], rb1[B_CHUNKS];
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
  }

  gemm_epilogue<BM, BN, WGM, WGN, ACT, RESID, GLU, CH>(p, acc, m0, n0, wm, wn_, lane);
}

template <int BM, int BN, int WGM, int WGN, int ACT, bool RESID, bool GLU, bool CH,
          int BKT>
int launch(const GemmArgs& a, const __bf16* Wh, hipStream_t stream) {
  const int tiles_m = cdiv(a.M, BM), tiles_n = cdiv(a.N, BN);
  const size_t lds = 2 * (BM + BN) * (BKT + 8) * sizeof(__bf16);
  auto kern = gemm_bf16s_kernel<BM, BN, WGM, WGN, ACT, RESID, GLU, CH, BKT>;
  WN_MAX_DYN_LDS(kern, lds);
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(WGM * WGN * 64), lds,
                     stream, a, Wh, tiles_m, tiles_n);
  WN_HIP(hipGetLastError());
  return 0;
}

// GLU_ONLY: as in gemm.hip -- only the kernels the tile rule below can reach are instantiated
template <int BM, int BN, int WGM, int WGN, int BKT, bool GLU_ONLY = false>
int dispatch_epi(const GemmArgs& a, const __bf16* Wh, bool ch, hipStream_t s) {
  const bool resid = a.resid != nullptr;
  if constexpr (GLU_ONLY) {
    static_assert(BN / WGN == 64, "GLU epilogue needs a 64-wide wave tile");
    if (a.glu) return launch<BM, BN, WGM, WGN, ACT_NONE, false, true, false, BKT>(a, Wh, s);
    set_error("gemm(bf16 stored): this block shape is built for the GLU epilogue only");
    return -1;
  } else {
  if (a.glu) {
    set_error("gemm(bf16 stored): GLU epilogue needs a 64-wide wave tile");
    return -1;
  }
  if (ch) {  // bf16 C: no residual (checked by the caller)
    switch (a.act) {
      case ACT_NONE: return launch<BM, BN, WGM, WGN, ACT_NONE, false, false, true, BKT>(a, Wh, s);
      case ACT_SILU: return launch<BM, BN, WGM, WGN, ACT_SILU, false, false, true, BKT>(a, Wh, s);
      case ACT_GELU: return launch<BM, BN, WGM, WGN, ACT_GELU, false, false, true, BKT>(a, Wh, s);
      case ACT_RELU: return launch<BM, BN, WGM, WGN, ACT_RELU, false, false, true, BKT>(a, Wh, s);
    }
  }
  switch (a.act) {
    case ACT_NONE:
      return resid ? launch<BM, BN, WGM, WGN, ACT_NONE, true, false, false, BKT>(a, Wh, s)
                   : launch<BM, BN, WGM, WGN, ACT_NONE, false, false, false, BKT>(a, Wh, s);
    case ACT_SILU:
      return resid ? launch<BM, BN, WGM, WGN, ACT_SILU, true, false, false, BKT>(a, Wh, s)
                   : launch<BM, BN, WGM, WGN, ACT_SILU, false, false, false, BKT>(a, Wh, s);
    case ACT_GELU:
      return resid ? launch<BM, BN, WGM, WGN, ACT_GELU, true, false, false, BKT>(a, Wh, s)
                   : launch<BM, BN, WGM, WGN, ACT_GELU, false, false, false, BKT>(a, Wh, s);
    case ACT_RELU:
      return resid ? launch<BM, BN, WGM, WGN, ACT_RELU, true, false, false, BKT>(a, Wh, s)
                   : launch<BM, BN, WGM, WGN, ACT_RELU, false, false, false, BKT>(a, Wh, s);
  }
  }
  set_error("gemm(bf16 stored): unsupported epilogue");
  return -1;
}

template <int BKT>
int dispatch_tile(const GemmArgs& a, const __bf16* Wh, bool ch, hipStream_t stream) {
  // the block-shape rule of gemm_bf16.hip (same tiles, same measurements)
  const int64_t t128 = (int64_t)cdiv(a.M, 128) * cdiv(a.N, 128);
  const int64_t t256 = (int64_t)cdiv(a.M, 256) * cdiv(a.N, 256);
  if (a.glu)
    return t128 >= 224 ? dispatch_epi<128, 128, 4, 2, BKT, true>(a, Wh, ch, stream)
                       : dispatch_epi<64, 128, 2, 2, BKT, true>(a, Wh, ch, stream);
  if constexpr (BKT == 64) {
    if (tune().gemm_tile_bf16 != 1 && t256 >= 256 && (t256 >= 1024 || a.K >= 2048))
      return dispatch_epi<256, 256, 4, 2, 64>(a, Wh, ch, stream);
  }
  return t128 >= 224 ? dispatch_epi<128, 128, 2, 4, BKT>(a, Wh, ch, stream)
                     : dispatch_epi<64, 64, 2, 2, BKT>(a, Wh, ch, stream);
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ y,
                                   int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 r;
    r[0] = (__bf16)v[0]; r[1] = (__bf16)v[1]; r[2] = (__bf16)v[2]; r[3] = (__bf16)v[3];
    *reinterpret_cast<bf16x4*>(y + i) = r;
  } else {
    for (int64_t j = i; j < n; ++j) y[j] = (__bf16)x[j];
  }
}

}  // namespace

int convert_f32_to_bf16(const float* x, void* y, int64_t n, hipStream_t s) {
  if (n <= 0) return 0;
  const int64_t thr = (n + 3) / 4;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)cdiv64(thr, 256)), dim3(256), 0,
                     s, x, reinterpret_cast<__bf16*>(y), n);
  WN_HIP(hipGetLastError());
  return 0;
}

// A: bf16 [M][lda] (a.A reinterpreted, lda in bf16 elements, % 8 == 0, 16-byte
// aligned base); Wh: the bf16 image of a.W; a.c_bf16: C is bf16 (ldc in elements).
int gemm_bf16_stored(const GemmArgs& a, const void* Wh, hipStream_t stream) {
  WN_CHECK(a.a_row_off == nullptr, "gemm(bf16 stored): gathered A is fp32 only");
  WN_CHECK(a.lda % 8 == 0, "gemm(bf16 stored): lda must be a multiple of 8 elements");
  WN_CHECK(!a.c_bf16 || (a.resid == nullptr && !a.glu),
           "gemm(bf16 stored): bf16 C without residual / GLU only");
  WN_CHECK((int64_t)a.M * a.lda * 2 < (int64_t(1) << 32) &&
               (int64_t)a.N * a.K * 2 < (int64_t(1) << 32),
           "gemm(bf16 stored): operand larger than the 32-bit byte offsets (4 GiB)");
  const __bf16* W = reinterpret_cast<const __bf16*>(Wh);
  // large shapes: the 256x256 direct-to-LDS pipelined kernel (gemm_bf16p.hip);
  // gemm_tile_bf16 = 8 forces it, any other forced tile keeps this file's kernels
  {
    const int64_t t256 = (int64_t)cdiv(a.M, 256) * cdiv(a.N, 256);
    const bool want = tune().gemm_tile_bf16 == 8 || (tune().gemm_tile_bf16 == 0 && t256 >= 192);
    if (want && gemm_bf16p_supported(a)) return gemm_bf16_pipelined(a, Wh, stream);
  }
  return a.K % 64 == 0 ? dispatch_tile<64>(a, W, a.c_bf16, stream)
                       : dispatch_tile<32>(a, W, a.c_bf16, stream);
}

}  // namespace wn
