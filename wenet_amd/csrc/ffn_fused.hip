// Fused feed-forward module in fp32: P[s] = act(X W1_s^T + b1_s) W2_s^T for the hidden
// slice s of the block, X = LayerNorm(x) -- PositionwiseFeedForward.forward,
// wenet/models/transformer/positionwise_feed_forward.py:50-58 (w_2(act(w_1 x))); the
// caller's next kernel adds the S slice partials, b2 and the residual and applies the
// following LayerNorm (encoder_layer.py:220-228,253-263).
//
// Why: the two FFN GEMMs are 46 % of the decode step (profiles/r02a) and each runs at
// ~0.6 of the fp32-MFMA rate; the (M, F) hidden tensor (65 MB at M = 7932, F = 2048)
// is written and re-read from HBM 24 times per step and every launch pays a prologue,
// an epilogue store burst and a partial last round.  Here the hidden tensor never
// leaves the CU: one block owns a 128-row tile of X and a slice of the hidden units and
// walks it in chunks of 64:
//   phase A   H[128 x 64]  = act(X[128 x D] W1[chunk]^T + b1)     K = D      (MFMA)
//             H -> LDS (the A operand of phase B, XOR-swizzled 256-B rows)
//   phase B   Y[128 x D]  += H[128 x 64] W2[:, chunk]^T           K = 64     (MFMA)
// with Y (64 x 64 ND fp32 per wave) resident in registers for the whole block.  The
// grid is tiles_m x S blocks with S chosen so that it fills the 256 CUs ONCE
// (62 x 4 = 248 at config 2): no tail round.  Exact fp32 arithmetic
// (v_mfma_f32_32x32x2_f32), the only change against the unfused path is that the sum
// over the hidden units is split into S partials (fp32 reordering, ~1e-7 relative).
//
// Operand stream: every operand tile is a 32-KB "stage" of 256 rows x 32 k (128 B per
// row) brought global -> LDS by DMA (buffer_load_dwordx4 ... lds), bank-swizzled on the
// source side like gemm_bf16p.hip; phase-A stages hold {X rows 0-127, W1 chunk rows
// 128-191}, phase-B stages 256 rows of W2.  A ring of 4 stages, ONE s_barrier per stage
// placed in the MIDDLE of the stage's MFMAs (see the loop), the DMA of a stage issued by
// one wave of each SIMD (alternating per stage) while the other keeps the matrix pipe fed,
// two stages in flight behind the stage being read; with 2048 .. 8192 MFMA cycles per stage the DMA latency
// is covered many times over.  No global loads
// other than the DMA inside the loop (the bias of the block's hidden slice is read into
// registers up front): hipcc would drain the DMA queue for them.
#include "common.h"
#include "gemm_epilogue.h"
#include "kernels.h"
#include "x6.h"

namespace wn {

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int FBM = 128;          // rows of X per block
constexpr int FHC = 64;           // hidden units per chunk
constexpr int STG = 32768;        // bytes per stage: 256 rows x 128 B
constexpr int HC_BYTES = FBM * FHC * 4;   // 32 KB

template <int ND, int ACT, int RING>
__global__ __launch_bounds__(512) void ffn_fused_kernel(FfnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_f[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  constexpr int D = 256 * ND;
  constexpr int NA = D / 32;            // phase-A stages per chunk
  constexpr int NB = (FHC / 32) * ND;   // phase-B stages per chunk
  constexpr int SPC = NA + NB;          // stages per chunk
  char* hc = smem_f + RING * STG;       // H chunk [128][64] fp32, swizzled

  const int tile_m = blockIdx.x / p.S, slice = blockIdx.x % p.S;
  const int m0 = tile_m * FBM;
  const int hs = p.F / p.S;             // hidden units of this block
  const int h_base = slice * hs;
  const int nchunk = hs / FHC;
  const int total = nchunk * SPC;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, li = lane & 31;

  // ---- DMA descriptors ---------------------------------------------------------------
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.X), 0, (int)min((int64_t)p.M * D * 4, (int64_t)0x7fffffff), 0x00020000);
  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.W1), 0, (int)min((int64_t)p.F * D * 4, (int64_t)0x7fffffff), 0x00020000);
  const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.W2), 0, (int)min((int64_t)D * p.F * 4, (int64_t)0x7fffffff), 0x00020000);
  // DMA issue is the job of ONE wave group per stage (waves 0-3 for even stages, 4-7 for
  // odd ones): a buffer_load ... lds costs the issuing wave 60-200 cycles, and with all 8
  // waves issuing right after the barrier both waves of every SIMD were busy with DMAs
  // instead of MFMAs.  Now the other wave of the SIMD can keep the pipe fed meanwhile
  // (measured neutral, 142 us either way at config 2, r02r: the issue time was not what
  // the matrix pipe waits for).  A wave of the issuing group moves 8 of the stage's
  // 32 one-KB pieces: piece pj = q*4 + (wave & 3) covers stage rows pj*8 .. +8; lane ->
  // row + (lane >> 3), LDS slot lane & 7, source slot = slot ^ ((row >> 1) & 7).
  const int w4 = wave & 3, grp = wave >> 2;
  unsigned vx[4], vw1[2], vw2[8];
  {
    const int rr = lane >> 3;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = (q * 4 + w4) * 8 + rr;              // stage row 0..255
      const unsigned swz = (unsigned)(((lane & 7) ^ ((r >> 1) & 7)) * 16);
      if (q < 4) vx[q] = (unsigned)min(m0 + r, p.M - 1) * (unsigned)(D * 4) + swz;
      else if (q < 6) vw1[q - 4] = (unsigned)(r - 128) * (unsigned)(D * 4) + swz;
      vw2[q] = (unsigned)r * (unsigned)p.F * 4u + swz;
    }
  }
  // stage g (clamped to the last one past the end: harmless re-load into a free buffer)
  auto issue = [&](int g) {
    if (grp != (g & 1)) return;
    g = min(g, total - 1);
    const int c = g / SPC, i = g - c * SPC;
    const int h0 = h_base + c * FHC;
    char* dst = smem_f + (g % RING) * STG + w4 * 1024;
    if (i < NA) {
      const int koff = i * 128;
      const int w1off = h0 * (D * 4) + koff;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(dst + q * 4096), 16, vx[q], koff,
                                                 0, 0);
      // rows 192-255 are not used in phase A; the W1 pieces again keep the DMA count per
      // stage uniform for the counted waits
#pragma unroll
      for (int q = 4; q < 8; ++q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_ptr)(dst + q * 4096), 16,
                                                 vw1[q & 1], w1off, 0, 0);
    } else {
      const int j = i - NA;
      const int kt = j / ND, nd = j - kt * ND;
      const int off = nd * 256 * p.F * 4 + (h0 + kt * 32) * 4;
#pragma unroll
      for (int q = 0; q < 8; ++q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (lds_ptr)(dst + q * 4096), 16, vw2[q], off,
                                                 0, 0);
    }
  };

  // ---- fragment addressing ---------------------------------------------------------------
  // stage rows: 128-B rows, 16-B slot (2 kk + hi) ^ ((row >> 1) & 7), kk = k-group of 8
  const int sw = (lane >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) foff[kk] = li * 128 + (((kk * 2 + hi) ^ sw) << 4);
  // phase A: 8 waves as 4 (M) x 2 (N), wave tile 32 x 32
  const int wmA = wave >> 1, wnA = wave & 1;
  // phase B: 2 (M) x 4 (N), wave tile 64 x 64 (per 256-column slab of Y)
  const int wmB = wave >> 2, wnB = wave & 3;
  // H chunk: 256-B rows, slot (k / 4) ^ (row & 15)
  const int hrow_sw = li & 15;

  // bias of the block's hidden slice: column (lane & 31) of this wave's phase-A tile
  constexpr int MAXC = 16;
  float b1v[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    b1v[c] = c < nchunk ? p.b1[h_base + c * FHC + wnA * 32 + li] : 0.f;
  // complete these loads BEFORE the first DMA is issued: a pending ordinary load would
  // make hipcc wait vmcnt(0) at its first use inside the loop and drain the DMA ring
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int c = 0; c < MAXC; ++c) asm volatile("" : "+v"(b1v[c]));

  f32x16 yacc[ND][2][2];
#pragma unroll
  for (int nd = 0; nd < ND; ++nd)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[nd][i][j][r] = 0.f;

  // ---- software pipeline ------------------------------------------------------------------
  // Invariant at the top of stage g: stages <= g have landed and are visible to every
  // wave, stages g+1, g+2 are in flight, and the k-groups 0, 1 of stage g sit in the X
  // fragment registers.  The ONE barrier of a stage sits in its MIDDLE: each wave first
  // waits (counted) for its own pieces of stage g+1, so past the barrier stage g+1 is
  // visible and every wave has left stage g-1 -- whose buffer the DMA of stage g+3 then
  // overwrites.  Fragments of the second half / the next stage are read before the MFMAs
  // of the first / second half are issued, so no MFMA ever waits for a ds_read behind a
  // barrier: the matrix pipe sees one uninterrupted stream.
  static_assert(RING == 4, "stage g is read while g+1 .. g+3 are in flight");
  struct FragA { f32x4 a, b; };
  struct FragB { f32x4 a[2], b[2]; };
  auto loadA = [&](int gg, int kk) {
    const char* st = smem_f + (gg % RING) * STG;
    FragA f;
    f.a = *reinterpret_cast<const f32x4*>(st + (wmA * 32) * 128 + foff[kk]);
    f.b = *reinterpret_cast<const f32x4*>(st + (128 + wnA * 32) * 128 + foff[kk]);
    return f;
  };
  auto loadB = [&](int gg, int kt, int kk) {
    const char* st = smem_f + (gg % RING) * STG;
    FragB f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = wmB * 64 + mt * 32 + li;
      const int k4 = kt * 8 + kk * 2 + hi;
      f.a[mt] = *reinterpret_cast<const f32x4*>(hc + row * 256 + ((k4 ^ hrow_sw) << 4));
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      f.b[nt] = *reinterpret_cast<const f32x4*>(st + (wnB * 64 + nt * 32) * 128 + foff[kk]);
    return f;
  };
  auto mid_barrier = [&](int gg) {
    // the group that issued stage gg+1 (two stages ago) waits for it; it has nothing else
    // in flight, and it is the group that issues stage gg+3 next
    if (grp == ((gg + 1) & 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue(gg + 3);
  };

  issue(0); issue(1); issue(2);
  if (grp == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // stage 0 (2 in flight)
  __builtin_amdgcn_s_barrier();
  FragA xa0 = loadA(0, 0), xa1 = loadA(0, 1);

  int g = 0;
  for (int c = 0; c < nchunk; ++c) {
    f32x16 hacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) hacc[r] = 0.f;
    auto mmaA = [&](const FragA& f) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
        hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[s], f.b[s], hacc, 0, 0, 0);
    };
    // ---- phase A -----------------------------------------------------------------------------
    for (int i = 0; i < NA; ++i, ++g) {
      const FragA ya0 = loadA(g, 2), ya1 = loadA(g, 3);
      mmaA(xa0); mmaA(xa1);
      mid_barrier(g);
      if (i + 1 < NA) { xa0 = loadA(g + 1, 0); xa1 = loadA(g + 1, 1); }
      mmaA(ya0); mmaA(ya1);
    }
    // ---- H = act(hacc + b1) -> LDS -------------------------------------------------------------
    {
      float bias = 0.f;
#pragma unroll
      for (int cc = 0; cc < MAXC; ++cc) bias = cc == c ? b1v[cc] : bias;
      const int col = wnA * 32 + li;                 // k of phase B
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float x = hacc[r] + bias;
        if (ACT == ACT_SILU) x = silu_fast(x);
        if (ACT == ACT_RELU) x = fmaxf(x, 0.f);
        if (ACT == ACT_GELU) x = 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
        const int row = wmA * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        *reinterpret_cast<float*>(hc + row * 256 + ((((col >> 2) ^ (row & 15))) << 4) +
                                  (col & 3) * 4) = x;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();       // H visible (the stage data already is)
    // ---- phase B -----------------------------------------------------------------------------
    FragB xb0 = loadB(g, 0, 0), xb1 = loadB(g, 0, 1);
#pragma unroll
    for (int j = 0; j < NB; ++j, ++g) {
      const int kt = j / ND, nd = j % ND;
      auto mmaB = [&](const FragB& f) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
              yacc[nd][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                  f.a[mt][s], f.b[nt][s], yacc[nd][mt][nt], 0, 0, 0);
      };
      const FragB yb0 = loadB(g, kt, 2), yb1 = loadB(g, kt, 3);
      mmaB(xb0); mmaB(xb1);
      mid_barrier(g);
      if (j + 1 < NB) {
        xb0 = loadB(g + 1, (j + 1) / ND, 0);
        xb1 = loadB(g + 1, (j + 1) / ND, 1);
      } else if (c + 1 < nchunk) {
        xa0 = loadA(g + 1, 0);
        xa1 = loadA(g + 1, 1);
      }
      mmaB(yb0); mmaB(yb1);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing (clamped) DMA

  // ---- partial Y of this hidden slice --------------------------------------------------------
  float* P = p.P + (int64_t)slice * p.M * D;
#pragma unroll
  for (int nd = 0; nd < ND; ++nd)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int col = nd * 256 + wnB * 64 + nt * 32 + li;
        const int row0 = m0 + wmB * 64 + mt * 32 + 4 * hi;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + (r & 3) + 8 * (r >> 2);
          if (row < p.M) P[(int64_t)row * D + col] = yacc[nd][mt][nt][r];
        }
      }
}


// x_new = x + alpha * (sum_s P[s] + b2); y = LN(x_new; w, b) [; y2 = LN(y; w2, b2)]
// One wave per row (RowRegs layout of encoder_kernels.hip: 4 consecutive columns per lane
// and 256-column chunk).  MODE 0: write x_new (in place) and y; MODE 1: x_new is not
// kept: x <- y = LN(x_new) and y2 = LN(y) (norm_final of a layer + the next layer's
// norm_ff_macaron, encoder_layer.py:263 -> :220); MODE 2: only x <- LN(x_new).
template <int E, int MODE>
__global__ __launch_bounds__(256) void ffn_reduce_ln_kernel(
    float* __restrict__ x, const float* __restrict__ P, int S, const float* __restrict__ b2,
    float alpha, const float* __restrict__ w, const float* __restrict__ b,
    const float* __restrict__ w2, const float* __restrict__ bb2, float* __restrict__ y,
    int M, float eps) {
  static_assert(E % 4 == 0, "4 columns per lane and chunk");
  constexpr int D = E * 64;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float v[E];
#pragma unroll
  for (int j = 0; j < E / 4; ++j) {
    const int c = j * 256 + lane * 4;
    f32x4 acc = *reinterpret_cast<const f32x4*>(b2 + c);
    // eight slice loads in flight per round, added in slice order (the same sum as a one-by-one
    // loop, without its S dependent memory round trips: S = 32 for a single utterance)
    int s = 0;
    for (; s + 8 <= S; s += 8) {
      f32x4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        t[u] = *reinterpret_cast<const f32x4*>(P + ((int64_t)(s + u) * M + row) * D + c);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += t[u];
    }
    for (; s < S; ++s)
      acc += *reinterpret_cast<const f32x4*>(P + ((int64_t)s * M + row) * D + c);
    const f32x4 xo = *reinterpret_cast<const f32x4*>(x + (int64_t)row * D + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[4 * j + e] = xo[e] + alpha * acc[e];
  }
  auto store = [&](float* dst) {
#pragma unroll
    for (int j = 0; j < E / 4; ++j)
      *reinterpret_cast<f32x4*>(dst + (int64_t)row * D + j * 256 + lane * 4) =
          f32x4{v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]};
  };
  auto norm = [&](const float* gw, const float* gb) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) s += v[e];
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {   // an explicit fma chain: the same bits in every kernel that
      const float d = v[e] - mean;  // forms this LayerNorm (gemm_x6r.hip PRO 1), whatever the
      q = __builtin_fmaf(d, d, q);  // vectoriser would make of d * d + q
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + eps);
#pragma unroll
    for (int j = 0; j < E / 4; ++j) {
      const f32x4 ww = *reinterpret_cast<const f32x4*>(gw + j * 256 + lane * 4);
      const f32x4 bv = *reinterpret_cast<const f32x4*>(gb + j * 256 + lane * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[4 * j + e] = (v[4 * j + e] - mean) * rstd * ww[e] + bv[e];
    }
  };
  if (MODE == 0) {
    store(x);
    norm(w, b);
    store(y);
  } else if (MODE == 1) {
    norm(w, b);
    store(x);
    norm(w2, bb2);
    store(y);
  } else {
    norm(w, b);
    store(x);
  }
}

// MODE 1 of ffn_reduce_ln_kernel with y2 = LN(y; w2, b2) leaving as its X3 PLANE IMAGE
// (x6.h: records [k block][row tile][plane] of 32 rows x 16 k, the operand order of the
// six-product kernels) instead of fp32 rows: the consumer -- the fused feed-forward module of the
// next layer, ffn_x6f.hip -- loads ready-made fragments.  The row arithmetic is the kernel
// above, statement for statement (a wave per row, the same sums in the same order: the image
// holds the exact split of the same fp32 values; tests/test_gpu_x6.py checks the bits).
// A block = 8 consecutive rows (two per wave): a (k block, half, plane) piece of the image is
// 16 bytes per row, so eight rows make one whole 128-byte line; the planes are turned through
// LDS and leave as 96 such lines per block.
template <int E>
__global__ __launch_bounds__(256) void ffn_reduce_ln_img_kernel(
    float* __restrict__ x, const float* __restrict__ P, int S, const float* __restrict__ b2,
    float alpha, const float* __restrict__ w, const float* __restrict__ b,
    const float* __restrict__ w2, const float* __restrict__ bb2, char* __restrict__ y3,
    int M, float eps) {
  static_assert(E % 4 == 0, "4 columns per lane and chunk");
  constexpr int D = E * 64;
  constexpr int NP = D / 8;                // 16-byte pieces of a row and plane: (k block, half)
  constexpr int SEG = 144;                 // LDS stride of a piece row (8 rows x 16 B + pad)
  __shared__ __attribute__((aligned(16))) char planes[3 * NP * SEG];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = blockIdx.x * 8;
  for (int rr = 0; rr < 2; ++rr) {
    const int rl = wave * 2 + rr;          // row inside the block
    const int row = m0 + rl;
    if (row >= M) continue;                // (wave-uniform)
    float v[E];
#pragma unroll
    for (int j = 0; j < E / 4; ++j) {
      const int c = j * 256 + lane * 4;
      f32x4 acc = *reinterpret_cast<const f32x4*>(b2 + c);
      int s = 0;
      for (; s + 8 <= S; s += 8) {
        f32x4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          t[u] = *reinterpret_cast<const f32x4*>(P + ((int64_t)(s + u) * M + row) * D + c);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += t[u];
      }
      for (; s < S; ++s)
        acc += *reinterpret_cast<const f32x4*>(P + ((int64_t)s * M + row) * D + c);
      const f32x4 xo = *reinterpret_cast<const f32x4*>(x + (int64_t)row * D + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[4 * j + e] = xo[e] + alpha * acc[e];
    }
    auto norm = [&](const float* gw, const float* gb) {
      float sm = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) sm += v[e];
      const float mean = wave_sum(sm) * (1.0f / D);
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float d = v[e] - mean;
        q = __builtin_fmaf(d, d, q);
      }
      const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + eps);
#pragma unroll
      for (int j = 0; j < E / 4; ++j) {
        const f32x4 ww = *reinterpret_cast<const f32x4*>(gw + j * 256 + lane * 4);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(gb + j * 256 + lane * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * j + e] = (v[4 * j + e] - mean) * rstd * ww[e] + bv[e];
      }
    };
    norm(w, b);
#pragma unroll
    for (int j = 0; j < E / 4; ++j)
      *reinterpret_cast<f32x4*>(x + (int64_t)row * D + j * 256 + lane * 4) =
          f32x4{v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]};
    norm(w2, bb2);
    // chunk j, lane l hold columns 256 j + 4 l .. + 3 = k block 16 j + l / 4, half (l / 2) & 1,
    // elements 4 (l & 1) .. + 3 of the half's eight
#pragma unroll
    for (int j = 0; j < E / 4; ++j) {
      bf16x4 h0, h1, h2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const Split3 sp = split3(v[4 * j + e]);
        h0[e] = sp.h0; h1[e] = sp.h1; h2[e] = sp.h2;
      }
      const int piece = (j * 16 + (lane >> 2)) * 2 + ((lane >> 1) & 1);   // (k block, half)
      char* o = planes + piece * SEG + rl * 16 + (lane & 1) * 8;
      *reinterpret_cast<bf16x4*>(o) = h0;
      *reinterpret_cast<bf16x4*>(o + NP * SEG) = h1;
      *reinterpret_cast<bf16x4*>(o + 2 * NP * SEG) = h2;
    }
  }
  __syncthreads();
  const int tiles = (M + 31) >> 5, tile = m0 >> 5, r32 = m0 & 31;
#pragma unroll
  for (int it = 0; it < 3 * NP * 8 / 256; ++it) {
    const int q = it * 256 + threadIdx.x;      // pieces of 16 B: (plane, k block, half, row)
    const int rl = q & 7, ph = q >> 3;         // ph = plane * NP + (k block * 2 + half)
    const int pl = ph / NP, kh = ph - pl * NP;
    if (m0 + rl < M) {
      const f32x4 val = *reinterpret_cast<const f32x4*>(planes + ph * SEG + rl * 16);
      *reinterpret_cast<f32x4*>(y3 + ((int64_t)(kh >> 1) * tiles + tile) * X3_TILE + pl * X3_REC +
                                (kh & 1) * 512 + (r32 + rl) * 16) = val;
    }
  }
}

}  // namespace


// hidden split: the largest S in {1, 2, 4, 8, 16} with tiles_m * S blocks filling the CUs
// once (128-row blocks, one per CU), at least one
// 64-wide chunk and at most 16 per block
int ffn_fused_split(int M, int D, int F) {
  const int tiles_m = cdiv(M, FBM), cap = 256;
  int S = 1;
  while (S < 16 && tiles_m * (S * 2) <= cap && F % (S * 2 * FHC) == 0) S *= 2;
  while (F / S / FHC > 16) {          // more chunks than the bias registers hold
    if (F % (S * 2 * FHC) != 0) return 0;
    S *= 2;
  }
  return S;
}

bool ffn_fused_supported(int M, int D, int F, int act) {
  if (!(D == 256 || D == 512) || F % FHC != 0 || M <= 0) return false;
  if (act != ACT_SILU && act != ACT_RELU && act != ACT_GELU) return false;
  if ((int64_t)M * D * 4 >= (int64_t(1) << 31) || (int64_t)F * D * 4 >= (int64_t(1) << 31))
    return false;
  const int S = ffn_fused_split(M, D, F);
  if (S <= 0) return false;
  // fewer blocks: the GEMM pair fills the chip better (tune().ffn_fused == 2: tests force it)
  return tune().ffn_fused == 2 || cdiv(M, FBM) * S >= 128;
}

template <int ND, int ACT>
int launch_ffn(const FfnArgs& a, hipStream_t s) {
  // 4 x 32 KB stages + 32 KB H chunk = the whole 160 KiB LDS of a CU
  const size_t lds = (size_t)4 * STG + HC_BYTES;
  const dim3 grid(cdiv(a.M, FBM) * a.S), blk(512);
  auto kern = ffn_fused_kernel<ND, ACT, 4>;
  WN_MAX_DYN_LDS(kern, lds);
  hipLaunchKernelGGL(kern, grid, blk, lds, s, a);
  WN_HIP(hipGetLastError());
  return 0;
}

int ffn_fused(const FfnArgs& a, hipStream_t s) {
  WN_CHECK(a.X && a.W1 && a.b1 && a.W2 && a.P && a.S > 0 && a.F % (a.S * FHC) == 0 &&
               a.F / a.S / FHC <= 16, "ffn_fused: bad arguments");
  WN_CHECK(a.D == 256 || a.D == 512, "ffn_fused: d_model 256 or 512");
#define WN_FFN(ND)                                                         \
  switch (a.act) {                                                         \
    case ACT_SILU: return launch_ffn<ND, ACT_SILU>(a, s);                  \
    case ACT_RELU: return launch_ffn<ND, ACT_RELU>(a, s);                  \
    case ACT_GELU: return launch_ffn<ND, ACT_GELU>(a, s);                  \
    default: break;                                                        \
  }
  if (a.D == 256) { WN_FFN(1) } else { WN_FFN(2) }
#undef WN_FFN
  set_error("ffn_fused: unsupported activation");
  return -1;
}

int ffn_reduce_ln(float* x, const float* P, int S, const float* b2, float alpha,
                  const float* w, const float* b, const float* w2, const float* bb2, float* y,
                  int M, int D, float eps, int mode, hipStream_t s) {
  WN_CHECK(M > 0 && (D == 256 || D == 512) && mode >= 0 && mode <= 2, "ffn_reduce_ln: shape");
  dim3 g(cdiv(M, 4)), t(256);
#define WN_RL(E, MODE)                                                                 \
  hipLaunchKernelGGL((ffn_reduce_ln_kernel<E, MODE>), g, t, 0, s, x, P, S, b2, alpha, w, b, \
                     w2, bb2, y, M, eps)
  if (D == 256) {
    if (mode == 0) WN_RL(4, 0); else if (mode == 1) WN_RL(4, 1); else WN_RL(4, 2);
  } else {
    if (mode == 0) WN_RL(8, 0); else if (mode == 1) WN_RL(8, 1); else WN_RL(8, 2);
  }
#undef WN_RL
  WN_HIP(hipGetLastError());
  return 0;
}

int ffn_reduce_ln_img(float* x, const float* P, int S, const float* b2, float alpha,
                      const float* w, const float* b, const float* w2, const float* bb2,
                      void* y3, int M, int D, float eps, hipStream_t s) {
  WN_CHECK(M > 0 && (D == 256 || D == 512) && y3 && w2 && bb2, "ffn_reduce_ln_img: shape");
  if (D == 256)
    hipLaunchKernelGGL(ffn_reduce_ln_img_kernel<4>, dim3(cdiv(M, 8)), dim3(256), 0, s, x, P, S,
                       b2, alpha, w, b, w2, bb2, reinterpret_cast<char*>(y3), M, eps);
  else
    hipLaunchKernelGGL(ffn_reduce_ln_img_kernel<8>, dim3(cdiv(M, 8)), dim3(256), 0, s, x, P, S,
                       b2, alpha, w, b, w2, bb2, reinterpret_cast<char*>(y3), M, eps);
  WN_HIP(hipGetLastError());
  return 0;
}

}  // namespace wn
