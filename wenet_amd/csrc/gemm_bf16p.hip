// Pipelined bf16 GEMM for the large shapes of the bf16 mode (Whisper-large widths):
// C = epi(A[M,K] * W[N,K]^T), A and W bf16 in HBM (the storage form of
// gemm_bf16s.hip), fp32 accumulate, C fp32 or bf16.  Same arithmetic as
// gemm_bf16s_kernel (k order inside an MFMA aside); what changes is the machine
// mapping, built for one 512-thread block per CU at 256 VGPRs:
//
//  * 256 x 256 block tile, K tile 64, 8 waves as 2 (M) x 4 (N), wave tile 128 x 64
//    = 8 accumulator tiles of v_mfma_f32_32x32x16_bf16 (128 VGPRs).
//  * operands go global -> LDS directly (buffer_load_dwordx4 ... lds, no staging
//    registers): a K tile is four 16-KB half-tiles {A rows 0-127, A rows 128-255,
//    W rows 0-127, W rows 128-255}, two K tiles of LDS (128 KB).  The LDS image of
//    a half-tile is [128 rows][8 x 16 B]; the DMA writes lane-linear, so the bank
//    swizzle slot' = slot ^ ((row >> 1) & 7) is applied to the SOURCE k-slot of
//    each lane and again on the fragment reads -- ds_read_b128 of 32 consecutive
//    rows at one k-slot then touches 16 distinct 16-B slots per service group.
//  * 4 phases per K tile; phase q computes one 64 x 32 quadrant of the wave tile
//    over the whole K tile (8 MFMAs) after reading only the fragments it is the
//    first to need (A0+B0 | B1 | A1 | -), and issues the DMA of one half-tile.
//    Half-tiles of K tile t+1 are issued in phases 4t-1 .. 4t+2 (one phase after
//    the last fragment read of K tile t-1, whose LDS they overwrite), waited for
//    with ONE counted s_waitcnt vmcnt(2) per K tile in phase 4t+3 -- the newest
//    half-tile stays in flight across the tile boundary -- and read from phase
//    4t+4 on.  Raw s_barrier only (a __syncthreads() would drain the DMA queue).
//  * the two waves of a SIMD (waves w and w+4) run one barrier apart: while one
//    is in its MFMA segment the other issues its ds_reads / DMA, so the matrix
//    pipe sees back-to-back MFMAs (MI355X_MICROARCH.md "Two waves per SIMD").
//  * operand-swapped MFMA (D = W_frag * A_frag^T): a lane owns ONE output row and
//    4 x 4 consecutive columns per 32 x 32 tile, so the epilogue issues 16-byte
//    (fp32 C) / 8-byte (bf16 C) stores and 16-byte residual loads.
//  * GELU in the epilogue uses the Abramowitz-Stegun 7.1.26 erf (|err| < 1.5e-7,
//    one v_exp + one v_rcp) instead of erff: the result is rounded to bf16 (2^-9
//    relative) or added into an fp32 stream that the next GEMM rounds to bf16.
#include "common.h"
#include "gemm_epilogue.h"

namespace wn {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int PBM = 256, PBN = 256, PBK = 64;
constexpr int HALF_BYTES = 128 * PBK * 2;     // 16 KB
constexpr int TILE_BYTES = 4 * HALF_BYTES;    // 64 KB: A lo, A hi, W lo, W hi

#define WN_LGKM0() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
                        __builtin_amdgcn_sched_barrier(0); } while (0)

__device__ __forceinline__ float gelu_as(float x) {
  // 0.5 x (1 + erf(x / sqrt 2)), erf by A&S 7.1.26 on |z|
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float e = 1.0f - poly * t * __expf(-z * z);   // erf(|z|)
  return 0.5f * x * (1.0f + copysignf(e, x));
}

// VAR: experiment / ablation bits (g_gemm_variant; only the plain fp32-C instantiation
// is built with VAR != 0): 1 no wave stagger, 2 no DMA in the main loop, 4 no fragment
// reads, 8 no MFMAs (2 / 4 / 8 give wrong results: timing ablations), 16 the
// issue order {A lo, A hi, W lo(t+2), W hi(t+2)} with two half-tiles in flight across
// the K-tile boundary.
template <int ACT, bool RESID, bool CH, int VAR = 0>
__global__ __launch_bounds__(512) void gemm_bf16p_kernel(
    GemmArgs p, const __bf16* __restrict__ Wh, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem_p[];
  typedef __attribute__((address_space(3))) void* lds_ptr;

  // ---- tile assignment: XCD-contiguous chunks, inside a chunk groups of 4 M
  // panels x all N tiles with M fastest: the 32 blocks an XCD runs at a time
  // share 4 A panels and 8 W panels ------------------------------------------
  const int nblk = tiles_m * tiles_n;
  const int bid = xcd_block_order(blockIdx.x, nblk);
  constexpr int GM = 4;
  const int per_group = GM * tiles_n;
  const int grp = bid / per_group;
  const int in_grp = bid - grp * per_group;
  const int gm = min(GM, tiles_m - grp * GM);   // rows of M panels in this group
  const int tm = grp * GM + in_grp % gm, tn = in_grp / gm;
  const int m0 = tm * PBM, n0 = tn * PBN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn_ = wave & 3;

  // ---- DMA descriptors -------------------------------------------------------
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.A), 0, (int)min((int64_t)p.M * p.lda * 2, (int64_t)0x7fffffff),
      0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(Wh), 0, (int)min((int64_t)p.N * p.K * 2, (int64_t)0x7fffffff),
      0x00020000);
  // per half-tile each thread moves 2 x 16 B: piece j covers rows (j*8 + wave)*8 ..
  // +8 of the half, lane -> (row = lane >> 3, LDS slot = lane & 7), source k-slot
  // = slot ^ ((row >> 1) & 7)
  unsigned va[2][2], vw[2][2];   // [half][piece] byte offsets of k tile 0
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = (j * 8 + wave) * 8 + (lane >> 3);
      const int ks = (lane & 7) ^ ((r >> 1) & 7);
      const int ar = min(m0 + h * 128 + r, p.M - 1);
      const int wr = min(n0 + h * 128 + r, p.N - 1);
      va[h][j] = ((unsigned)ar * (unsigned)p.lda + (unsigned)ks * 8u) * 2u;
      vw[h][j] = ((unsigned)wr * (unsigned)p.K + (unsigned)ks * 8u) * 2u;
    }
  const int nk = p.K / PBK;

  // stage half-tile `half` (0,1 = A lo/hi; 2,3 = W lo/hi) of K tile `kt` into LDS
  // tile buffer `par`
  auto stage = [&](int kt, int half, int par, bool prologue = false) {
    if ((VAR & 2) && !prologue) return;
    const int ktc = min(kt, nk - 1);         // past the end: harmless re-load
    const int soff = ktc * (PBK * 2);
    char* dst = smem_p + par * TILE_BYTES + half * HALF_BYTES + wave * 1024;
    if (half < 2) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)dst, 16, va[half][0], soff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(dst + 8192), 16, va[half][1],
                                               soff, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)dst, 16, vw[half - 2][0], soff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(dst + 8192), 16, vw[half - 2][1],
                                               soff, 0, 0);
    }
  };

  // ---- fragment addresses ------------------------------------------------------
  // row (lane & 31) of a 32-row block, k-step ks: slot (2 ks + (lane >> 5)) ^ sw
  const int sw = (lane >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    foff[ks] = (lane & 31) * 128 + (((ks * 2 + (lane >> 5)) ^ sw) << 4);
  const char* a_base = smem_p + wm * HALF_BYTES;                       // A half = wm
  const char* b_base = smem_p + (2 + (wn_ >> 1)) * HALF_BYTES + (wn_ & 1) * (64 * 128);

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  bf16x8 fa[2][4], fb0[4], fb1[4];
  auto read_a = [&](int par, int qi) {
    if (VAR & 4) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(fa[rb][ks]));
      return;
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        fa[rb][ks] = *reinterpret_cast<const bf16x8*>(
            a_base + par * TILE_BYTES + (qi * 64 + rb * 32) * 128 + foff[ks]);
  };
  auto read_b = [&](int par, int qj, bf16x8 (&fb)[4]) {
    if (VAR & 4) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(fb[ks]));
      return;
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      fb[ks] = *reinterpret_cast<const bf16x8*>(b_base + par * TILE_BYTES + qj * 32 * 128 +
                                                foff[ks]);
  };
  auto mma = [&](int qi, int qj, const bf16x8 (&fb)[4]) {
    if (VAR & 8) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        asm volatile("" :: "v"(fb[ks]));
        asm volatile("" :: "v"(fa[0][ks]));
        asm volatile("" :: "v"(fa[1][ks]));
      }
      return;
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
        acc[2 * qi + rb][qj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
            fb[ks], fa[rb][ks], acc[2 * qi + rb][qj], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  // one K tile = 4 phases; `par` is the tile's LDS buffer (compile-time in the
  // unrolled callers)
  auto ktile = [&](int t, int par) {
    if constexpr ((VAR & 16) != 0) {
      // issue order with TWO half-tiles in flight across the K-tile boundary:
      // tile t+1 = {W lo @4t-2, W hi @4t-1, A lo @4t, A hi @4t+1}
      read_a(par, 0);
      read_b(par, 0, fb0);
      stage(t + 1, 0, par ^ 1);
      __builtin_amdgcn_s_barrier();
      WN_LGKM0();
      mma(0, 0, fb0);
      __builtin_amdgcn_s_barrier();
      read_b(par, 1, fb1);           // last read of this tile's W halves
      stage(t + 1, 1, par ^ 1);
      WN_LGKM0();
      __builtin_amdgcn_s_barrier();
      mma(0, 1, fb1);
      __builtin_amdgcn_s_barrier();
      read_a(par, 1);                // last read of this tile's A halves
      stage(t + 2, 2, par);
      WN_LGKM0();
      __builtin_amdgcn_s_barrier();
      mma(1, 1, fb1);
      __builtin_amdgcn_s_barrier();
      stage(t + 2, 3, par);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      mma(1, 0, fb0);
      __builtin_amdgcn_s_barrier();
      return;
    }
    // phase 0: A rows 0-63 + W cols 0-31; DMA half 1 of tile t+1
    read_a(par, 0);
    read_b(par, 0, fb0);
    stage(t + 1, 1, par ^ 1);
    __builtin_amdgcn_s_barrier();
    WN_LGKM0();
    mma(0, 0, fb0);
    __builtin_amdgcn_s_barrier();
    // phase 1: W cols 32-63; DMA half 2 of tile t+1
    read_b(par, 1, fb1);
    stage(t + 1, 2, par ^ 1);
    __builtin_amdgcn_s_barrier();
    WN_LGKM0();
    mma(0, 1, fb1);
    __builtin_amdgcn_s_barrier();
    // phase 2: A rows 64-127 -- the LAST read of this tile's buffer, retired before
    // the barrier because the next phase's DMA overwrites it; DMA half 3 of t+1
    read_a(par, 1);
    stage(t + 1, 3, par ^ 1);
    WN_LGKM0();
    __builtin_amdgcn_s_barrier();
    mma(1, 1, fb1);
    __builtin_amdgcn_s_barrier();
    // phase 3: no reads; DMA half 0 of tile t+2 into THIS tile's buffer; everything
    // but that newest half-tile must have landed before tile t+1 is read
    stage(t + 2, 0, par);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    mma(1, 0, fb0);
    __builtin_amdgcn_s_barrier();
  };

  // ---- prologue: K tile 0 and the first half-tile of K tile 1 ----------------------
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      fa[rb][ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      fb0[ks] = fa[rb][ks];
      fb1[ks] = fa[rb][ks];
    }
  stage(0, 0, 0, true); stage(0, 1, 0, true); stage(0, 2, 0, true); stage(0, 3, 0, true);
  if constexpr ((VAR & 16) != 0) {
    stage(1, 2, 1, true); stage(1, 3, 1, true);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    stage(1, 0, 1, true);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  // the second wave of each SIMD runs one barrier behind the first
  if (!(VAR & 1) && wm == 1) __builtin_amdgcn_s_barrier();
  int t = 0;
  for (; t + 1 < nk; t += 2) {
    ktile(t, 0);
    ktile(t + 1, 1);
  }
  if (t < nk) ktile(t, 0);
  if (!(VAR & 1) && wm == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing (clamped) DMA

  // ---- epilogue: lane = output row, 4 consecutive columns per register quad ------
  const int hi4 = (lane >> 5) * 4;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int cb = n0 + wn_ * 64 + nb * 32 + hi4;     // + 8 g + (0..3)
    f32x4 bias4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = cb + 8 * g;
      bias4[g] = (p.bias && c < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + c)
                                      : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const int row = m0 + wm * 128 + mb * 32 + (lane & 31);
      if (row >= p.M) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = cb + 8 * g;
        if (c >= p.N) continue;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = acc[mb][nb][4 * g + e] + bias4[g][e];
          if (ACT == ACT_SILU) x = silu_fast(x);
          if (ACT == ACT_RELU) x = fmaxf(x, 0.0f);
          if (ACT == ACT_GELU) x = gelu_as(x);
          v[e] = x * p.alpha;
        }
        if constexpr (CH) {
          bf16x4 h;
          h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
          *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.C) + (int64_t)row * p.ldc + c) = h;
        } else {
          if constexpr (RESID)
            v += *reinterpret_cast<const f32x4*>(p.resid + (int64_t)row * p.ldr + c);
          *reinterpret_cast<f32x4*>(p.C + (int64_t)row * p.ldc + c) = v;
        }
      }
    }
  }
}

template <int ACT, bool RESID, bool CH, int VAR = 0>
int launch_p(const GemmArgs& a, const __bf16* Wh, hipStream_t stream) {
  const int tiles_m = cdiv(a.M, PBM), tiles_n = cdiv(a.N, PBN);
  const size_t lds = 2 * TILE_BYTES;
  auto kern = gemm_bf16p_kernel<ACT, RESID, CH, VAR>;
  static bool attr_done = false;  // per instantiation
  if (!attr_done) {
    WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), lds, stream, a, Wh, tiles_m,
                     tiles_n);
  WN_HIP(hipGetLastError());
  return 0;
}

}  // namespace

// Shapes the pipelined kernel takes (everything else stays on gemm_bf16s_kernel)
bool gemm_bf16p_supported(const GemmArgs& a) {
  return !a.glu && a.a_row_off == nullptr && a.K % PBK == 0 && a.K >= 2 * PBK &&
         a.N % 8 == 0 && a.lda % 8 == 0 && a.ldc % 4 == 0 &&
         (a.resid == nullptr || a.ldr % 4 == 0) && !(a.c_bf16 && a.resid) &&
         (int64_t)a.M * a.lda * 2 < (int64_t(1) << 31) &&
         (int64_t)a.N * a.K * 2 < (int64_t(1) << 31);
}

int gemm_bf16_pipelined(const GemmArgs& a, const void* Wh, hipStream_t stream) {
  WN_CHECK(gemm_bf16p_supported(a), "gemm(bf16 pipelined): unsupported shape");
  const __bf16* W = reinterpret_cast<const __bf16*>(Wh);
  const bool resid = a.resid != nullptr;
  if (g_gemm_variant != 0 && !a.c_bf16 && !resid && a.act == ACT_NONE) {
    switch (g_gemm_variant) {   // experiments (tools/bench_gemm.py --variants)
      case 1: return launch_p<ACT_NONE, false, false, 1>(a, W, stream);
      case 2: return launch_p<ACT_NONE, false, false, 2>(a, W, stream);
      case 4: return launch_p<ACT_NONE, false, false, 4>(a, W, stream);
      case 6: return launch_p<ACT_NONE, false, false, 6>(a, W, stream);
      case 8: return launch_p<ACT_NONE, false, false, 8>(a, W, stream);
      case 10: return launch_p<ACT_NONE, false, false, 10>(a, W, stream);
      case 16: return launch_p<ACT_NONE, false, false, 16>(a, W, stream);
      case 17: return launch_p<ACT_NONE, false, false, 17>(a, W, stream);
      default: break;
    }
  }
  if (a.c_bf16) {
    switch (a.act) {
      case ACT_NONE: return launch_p<ACT_NONE, false, true>(a, W, stream);
      case ACT_SILU: return launch_p<ACT_SILU, false, true>(a, W, stream);
      case ACT_GELU: return launch_p<ACT_GELU, false, true>(a, W, stream);
      case ACT_RELU: return launch_p<ACT_RELU, false, true>(a, W, stream);
    }
  }
  switch (a.act) {
    case ACT_NONE: return resid ? launch_p<ACT_NONE, true, false>(a, W, stream)
                                : launch_p<ACT_NONE, false, false>(a, W, stream);
    case ACT_SILU: return resid ? launch_p<ACT_SILU, true, false>(a, W, stream)
                                : launch_p<ACT_SILU, false, false>(a, W, stream);
    case ACT_GELU: return resid ? launch_p<ACT_GELU, true, false>(a, W, stream)
                                : launch_p<ACT_GELU, false, false>(a, W, stream);
    case ACT_RELU: return resid ? launch_p<ACT_RELU, true, false>(a, W, stream)
                                : launch_p<ACT_RELU, false, false>(a, W, stream);
  }
  set_error("gemm(bf16 pipelined): unsupported epilogue");
  return -1;
}

}  // namespace wn
