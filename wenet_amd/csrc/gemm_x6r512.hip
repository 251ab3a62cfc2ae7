// Row-block fp32 GEMM on the bf16 matrix cores (six plane products) for the encoder's small
// projections at d_model = 512 -- K = 512, N = 512 .. 1536: the d = 512 counterpart of
// gemm_x6r.hip (round 4).  Same launches per Conformer layer as at d = 256:
//   * QKV projection (attention.py:109-131; its input LN_mha(x + 0.5 FFN_macaron) still comes
//     from ffn_reduce_ln);
//   * attention output projection + residual + LN_conv chained with pointwise_conv1 + GLU
//     (attention.py:176, encoder_layer.py:238-240, convolution.py:115-118);
//   * pointwise_conv2 + residual + LN_ff (convolution.py:148, encoder_layer.py:251-255).
// Before (profiles/r06b_kernel_stats_config4_streams1.md, per layer at config 4): QKV 76 us on
// the tile GEMM + 7 us plane-split pass + 18 us ffn_reduce_ln; out-projection and
// pointwise_conv2 49 us each on v_mfma_f32 + 7 us LayerNorm each; pointwise_conv1 + GLU 84 us on
// v_mfma_f32 -- 298 us of a 1.1-ms layer for 14 % of its FLOPs.
//
// What differs from the K = 256 kernel: 32 rows x 512 k x three planes are 96 KB -- 384 registers
// per lane, too many next to the accumulators and the W fragments.  The block splits its rows
// ONCE into an X3-format image in LDS ([k block][plane][lane x 16 B]: exactly the fragment a
// lane needs, conflict-free ds_read_b128), each wave splitting a quarter of the k blocks, and
// every wave reads its "B" operand fragments from there (3 x 16 B per k block, one k block
// ahead).  Everything else is gemm_x6r.hip: one wave per SIMD, wave w owns the columns
// [128 w, 128 w + 128) of a 512-column pass (N = 1536: three passes over the same X image), W
// fragments straight from the weight plane image in L2 (PF k blocks ahead), lane = row in the
// accumulators, LayerNorm statistics across the four waves through LDS, every global access as
// contiguous row segments through wave-private LDS patches.
// LDS: 96 KB X image + 36 KB (fp32 half rows of the prologue / GLU patches); the epilogue's
// wave patches (4 x 16.5 KB) reuse the X image once the MFMA loop is done.
#include "common.h"
#include "kernels.h"
#include "x6.h"

namespace wn {

namespace {

constexpr int K5 = 512;
constexpr int KB5 = K5 / 16;            // 32 k blocks
constexpr int XIMG = KB5 * X3_TILE;     // 96 KB
constexpr int HPATCH = 36 * 1024;       // fp32 half rows [32][1040 B] / GLU wave patches
constexpr int NT5 = 4;                  // 32-column tiles per wave and pass
constexpr int PF5 = 2;                  // W fragment prefetch distance (k blocks)

// EPI 0: C = acc + bias (N = 512 passes); EPI 1: x_out = resid + alpha (acc + bias),
// y = LayerNorm(x_out) (N = 512); EPI 3: EPI 1, then C = GLU(y W3b^T + bias2) with W3b the image
// of a 1024 x 512 weight whose rows are permuted per 64 as [32 values | 32 gates] (y itself is
// stored only if p.y is set).  (A prologue that formed the rows from the feed-forward slice
// partials, like gemm_x6r.hip's PRO, was built and measured in round 4: 107 us against 79 us +
// 20 us for the separate ffn_reduce_ln launch at config 4 -- removed.)
template <int EPI>
__global__ __launch_bounds__(256, 1) void x6r512_kernel(X6RArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem5[];
  __shared__ float red[2][4][32];
  char* ximg = smem5;
  char* hpatch = smem5 + XIMG;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < 4);
  const int hi = lane >> 5, li = lane & 31;
  const int m0 = blockIdx.x * 32;
  const int Tn = (p.N + 31) >> 5;
  const int64_t kstride = (int64_t)Tn * X3_TILE;

  // W fragments of pass ps: records [k block][tile][plane], this wave's tiles 16 ps + 4 w ..
  bf16x8 wf[PF5 + 1][NT5][3];
  auto w_base = [&](const void* W3, int ps) {
    return reinterpret_cast<const char*>(W3) + ((int64_t)(ps * 16 + wave * NT5) * 3) * X3_REC +
           lane * 16;
  };
  auto load_w = [&](const char* wb, int64_t kst, int ks) {
    const char* q = wb + ks * kst;
#pragma unroll
    for (int t = 0; t < NT5; ++t)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        wf[ks % (PF5 + 1)][t][pl] = *reinterpret_cast<const bf16x8*>(q + (t * 3 + pl) * X3_REC);
  };

  // ---- prologue: the block's 32 rows -> X image in LDS ---------------------------------------
  // rows as fp32, whole rows per instruction (wave w: rows 8 w .. 8 w + 7, two 1-KB halves)
  f32x4 rowv[2][8];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = min(m0 + wave * 8 + j, p.M - 1);
      rowv[hh][j] =
          *reinterpret_cast<const f32x4*>(p.A + (int64_t)r * p.lda + hh * 256 + lane * 4);
    }
  const char* wb = w_base(p.W3, 0);
#pragma unroll
  for (int s = 0; s < PF5; ++s) load_w(wb, kstride, s);

  // fp32 half rows -> LDS (row stride 1040 B: conflict-free for the lane = row reads) -> this
  // wave's quarter of the half's k blocks in the fragment layout (lane = row li, k half hi) ->
  // exact three-way split -> X image.  `src`: half hh of row 8 w + j for lane.
  auto image_half = [&](int hh, const f32x4 (&src)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      *reinterpret_cast<f32x4*>(hpatch + (wave * 8 + j) * 1040 + lane * 16) = src[j];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ks = wave * 4 + u;              // k block of this half
      const f32x4 xa = *reinterpret_cast<const f32x4*>(hpatch + li * 1040 + ks * 64 + hi * 32);
      const f32x4 xb =
          *reinterpret_cast<const f32x4*>(hpatch + li * 1040 + ks * 64 + hi * 32 + 16);
      bf16x8 x0, x1, x2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const Split3 sa = split3(xa[e]), sb = split3(xb[e]);
        x0[e] = sa.h0; x1[e] = sa.h1; x2[e] = sa.h2;
        x0[4 + e] = sb.h0; x1[4 + e] = sb.h1; x2[4 + e] = sb.h2;
      }
      char* o = ximg + (int64_t)((hh * 16 + ks) * 3) * X3_REC + lane * 16;
      *reinterpret_cast<bf16x8*>(o) = x0;
      *reinterpret_cast<bf16x8*>(o + X3_REC) = x1;
      *reinterpret_cast<bf16x8*>(o + 2 * X3_REC) = x2;
    }
    __syncthreads();                            // half patch reusable, image half visible
  };
  image_half(0, rowv[0]);
  image_half(1, rowv[1]);

  // ---- one 512-column pass: acc[t] = W tile (16 ps + 4 w + t) x X^T ---------------------------
  // plane products, the small ones first: (W plane, activation plane); the tiles alternate so
  // that no MFMA waits for its predecessor's accumulator
  constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};
  f32x16 acc[NT5];
  auto gemm_pass = [&](const char* wbp, int64_t kst) {   // (the first PF5 W loads are issued)
#pragma unroll
    for (int t = 0; t < NT5; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    bf16x8 xf[2][3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      xf[0][pl] = *reinterpret_cast<const bf16x8*>(ximg + pl * X3_REC + lane * 16);
#pragma unroll
    for (int ks = 0; ks < KB5; ++ks) {
      if (ks + PF5 < KB5) load_w(wbp, kst, ks + PF5);
      if (ks + 1 < KB5) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          xf[(ks + 1) & 1][pl] = *reinterpret_cast<const bf16x8*>(
              ximg + (int64_t)((ks + 1) * 3 + pl) * X3_REC + lane * 16);
      }
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int t = 0; t < NT5; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks % (PF5 + 1)][t][PW[q]],
                                                           xf[ks & 1][PX[q]], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- epilogue helpers: lane = row, registers = columns col0 + 32 t + 8 g + 4 hi + e ----------
  // every tile of C / x_out / y and of the residual goes through a wave-private LDS patch (32
  // rows x 512 B, row stride + 16 B) and crosses the memory pipe as 512-B row segments
  constexpr int WCOLS = NT5 * 32;                  // columns a wave owns in a pass
  constexpr int SEG = NT5 * 128, PST = SEG + 16, LPR = SEG / 16, NIT = 32 * LPR / 64;   // (bytes)
  char* wp = ximg + wave * (32 * PST);          // (only once the X image is dead)
  auto put = [&](const f32x4 (&v)[NT5][4]) {
#pragma unroll
    for (int t = 0; t < NT5; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(wp + li * PST + (t * 32 + 8 * g + 4 * hi) * 4) = v[t][g];
  };
  auto get = [&](f32x4 (&v)[NT5][4]) {
#pragma unroll
    for (int t = 0; t < NT5; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        v[t][g] = *reinterpret_cast<const f32x4*>(wp + li * PST + (t * 32 + 8 * g + 4 * hi) * 4);
  };
  auto store_rows = [&](float* base, int ld, int col0) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = it * 64 + lane, r = q / LPR, pc = q - r * LPR;
      const f32x4 v = *reinterpret_cast<const f32x4*>(wp + r * PST + pc * 16);
      if (m0 + r < p.M && col0 + pc * 4 < p.N)
        *reinterpret_cast<f32x4*>(base + (int64_t)(m0 + r) * ld + col0 + pc * 4) = v;
    }
  };

  if constexpr (EPI == 0) {
    // N / 512 passes over the same X image, which therefore stays alive: the C tiles leave
    // through the half-row area instead (36 KB: 4 x 8.5 KB), two tiles of a wave at a time
    constexpr int HSEG = 2 * 128, HPST = HSEG + 16, HLPR = HSEG / 16, HNIT = 32 * HLPR / 64;
    char* hp = hpatch + wave * (32 * HPST);     // 4 x 8.5 KB = 34 KB
    const int npass = p.N / 512;
    for (int ps = 0; ps < npass; ++ps) {
      // the pass's bias vectors as ONE batch of loads under the GEMM (inside the store loops,
      // behind `if (bias)`, every load was waited for on its own: gemm_x6r.hip, r07v)
      f32x4 eb[NT5][4];
#pragma unroll
      for (int t = 0; t < NT5; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          eb[t][g] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (p.bias) eb[t][g] = *reinterpret_cast<const f32x4*>(p.bias + ps * 512 + wave * WCOLS + t * 32 + 8 * g + 4 * hi);
        }
      const char* wbp = w_base(p.W3, ps);
      gemm_pass(wbp, kstride);
      // the next pass's first W fragments go out BEFORE this pass's stores (vmcnt retires in
      // order: loads queued behind the stores would wait for the whole burst to drain)
      if (ps + 1 < npass) {
        const char* wbn = w_base(p.W3, ps + 1);
#pragma unroll
        for (int s = 0; s < PF5; ++s) load_w(wbn, kstride, s);
      }
      const int col0 = ps * 512 + wave * WCOLS;
#pragma unroll
      for (int u = 0; u < 2; ++u) {             // tiles 2 u, 2 u + 1
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int t = 2 * u + tt;
            const f32x4 v = f32x4{acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2],
                                  acc[t][4 * g + 3]} + eb[t][g];
            *reinterpret_cast<f32x4*>(hp + li * HPST + (tt * 32 + 8 * g + 4 * hi) * 4) = v;
          }
#pragma unroll
        for (int it = 0; it < HNIT; ++it) {
          const int q = it * 64 + lane, r = q / HLPR, pc = q - r * HLPR;
          const f32x4 v = *reinterpret_cast<const f32x4*>(hp + r * HPST + pc * 16);
          if (m0 + r < p.M)
            *reinterpret_cast<f32x4*>(p.C + (int64_t)(m0 + r) * p.ldc + col0 + u * 64 + pc * 4) = v;
        }
      }
    }
  } else {
    // the pass's bias vectors as ONE batch of loads under the GEMM (inside the store loops,
    // behind `if (bias)`, every load was waited for on its own: gemm_x6r.hip, r07v)
    f32x4 eb[NT5][4];
#pragma unroll
    for (int t = 0; t < NT5; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        eb[t][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) eb[t][g] = *reinterpret_cast<const f32x4*>(p.bias + wave * WCOLS + t * 32 + 8 * g + 4 * hi);
      }
    gemm_pass(wb, kstride);
    const int col0 = wave * WCOLS;
    __syncthreads();                            // the X image is dead: wave patches
    f32x4 v[NT5][4], rs[NT5][4];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = it * 64 + lane, r = q / LPR, pc = q - r * LPR;
      const int rc = min(m0 + r, p.M - 1);
      *reinterpret_cast<f32x4*>(wp + r * PST + pc * 16) =
          *reinterpret_cast<const f32x4*>(p.resid + (int64_t)rc * p.ldr + col0 + pc * 4);
    }
    get(rs);
    float s1 = 0.f;
#pragma unroll
    for (int t = 0; t < NT5; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 a = f32x4{acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2],
                              acc[t][4 * g + 3]} + eb[t][g];
        v[t][g] = rs[t][g] + p.alpha * a;
        s1 += (v[t][g][0] + v[t][g][1]) + (v[t][g][2] + v[t][g][3]);
      }
    put(v);
    store_rows(p.x_out, p.ldx, col0);
    // mean over the row's 512 columns: lane pair, then the four waves
    s1 += __shfl_xor(s1, 32, 64);
    if (hi == 0) red[0][wave][li] = s1;
    __syncthreads();
    const float mean =
        ((red[0][0][li] + red[0][1][li]) + (red[0][2][li] + red[0][3][li])) * (1.0f / K5);
    float s2 = 0.f;
#pragma unroll
    for (int t = 0; t < NT5; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[t][g][e] - mean;
          s2 += d * d;
        }
    s2 += __shfl_xor(s2, 32, 64);
    if (hi == 0) red[1][wave][li] = s2;
    __syncthreads();
    const float var =
        ((red[1][0][li] + red[1][1][li]) + (red[1][2][li] + red[1][3][li])) * (1.0f / K5);
    const float rstd = 1.0f / sqrtf(var + p.eps);
#pragma unroll
    for (int t = 0; t < NT5; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = col0 + t * 32 + 8 * g + 4 * hi;
        const f32x4 w = *reinterpret_cast<const f32x4*>(p.ln_w + c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.ln_b + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[t][g][e] = (v[t][g][e] - mean) * rstd * w[e] + b[e];
      }
    if (p.y != nullptr) {
      put(v);
      store_rows(p.y, p.ldy, col0);
    }
    if constexpr (EPI == 1) {
      if (p.y3 != nullptr) {      // y as its X3 plane image (x6.h x3_store_tile)
        const int tiles_m = (p.M + 31) >> 5;
#pragma unroll
        for (int t = 0; t < NT5; ++t)
          x3_store_tile(v[t], p.y3, (col0 + t * 32) >> 4, tiles_m, m0 >> 5, hi, li);
      }
    }
    if constexpr (EPI == 3) {
      // ---- chained: C = GLU(y W3b^T + bias2), N2 = 1024: the LayerNorm rows never leave the CU.
      // Half by half (waves 2 hh, 2 hh + 1 own the columns of k half hh): fp32 half rows into
      // the half patch, every wave splits its quarter of the half's k blocks into the X image
      const char* wb2 = w_base(p.W3b, 0);
      const int64_t kst2 = (int64_t)32 * X3_TILE;        // 1024 / 32 tiles per k block
#pragma unroll
      for (int s = 0; s < PF5; ++s) load_w(wb2, kst2, s);
      __syncthreads();                          // the wave patches (in the image area) are dead
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        if ((wave >> 1) == hh) {
#pragma unroll
          for (int t = 0; t < NT5; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              *reinterpret_cast<f32x4*>(hpatch + li * 1040 +
                                        ((wave & 1) * WCOLS + t * 32 + 8 * g + 4 * hi) * 4) =
                  v[t][g];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int ks = wave * 4 + u;
          const f32x4 xa = *reinterpret_cast<const f32x4*>(hpatch + li * 1040 + ks * 64 + hi * 32);
          const f32x4 xb =
              *reinterpret_cast<const f32x4*>(hpatch + li * 1040 + ks * 64 + hi * 32 + 16);
          bf16x8 x0, x1, x2;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const Split3 sa = split3(xa[e]), sb = split3(xb[e]);
            x0[e] = sa.h0; x1[e] = sa.h1; x2[e] = sa.h2;
            x0[4 + e] = sb.h0; x1[4 + e] = sb.h1; x2[4 + e] = sb.h2;
          }
          char* o = ximg + (int64_t)((hh * 16 + ks) * 3) * X3_REC + lane * 16;
          *reinterpret_cast<bf16x8*>(o) = x0;
          *reinterpret_cast<bf16x8*>(o + X3_REC) = x1;
          *reinterpret_cast<bf16x8*>(o + 2 * X3_REC) = x2;
        }
        __syncthreads();
      }
      // two passes of 512 image columns = 256 GLU columns each; wave w: value / gate tile pairs
      // (4 w, 4 w + 1), (4 w + 2, 4 w + 3) of the pass
      char* wp2 = hpatch + wave * (32 * 272);   // 32 rows x (2 x 128 B + 16): 4 x 8.5 KB
      for (int ps = 0; ps < 2; ++ps) {
        // the pass's bias vectors as ONE batch of loads under the GEMM (inside the store loops,
        // behind `if (bias)`, every load was waited for on its own: gemm_x6r.hip, r07v)
        f32x4 eb2[NT5][4];
#pragma unroll
        for (int t = 0; t < NT5; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            eb2[t][g] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias2) eb2[t][g] = *reinterpret_cast<const f32x4*>(p.bias2 + ps * 512 + wave * WCOLS + t * 32 + 8 * g + 4 * hi);
          }
        const char* wbp = w_base(p.W3b, ps);
        gemm_pass(wbp, kst2);
        if (ps == 0) {                          // (ahead of the stores, see EPI 0)
          const char* wbn = w_base(p.W3b, 1);
#pragma unroll
          for (int s = 0; s < PF5; ++s) load_w(wbn, kst2, s);
        }
        const int c2 = ps * 512 + wave * WCOLS; // first column of the wave's tiles in the image
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int c = c2 + 2 * u * 32 + 8 * g + 4 * hi;
            f32x4 a = f32x4{acc[2 * u][4 * g], acc[2 * u][4 * g + 1], acc[2 * u][4 * g + 2],
                            acc[2 * u][4 * g + 3]};
            f32x4 gt = f32x4{acc[2 * u + 1][4 * g], acc[2 * u + 1][4 * g + 1],
                             acc[2 * u + 1][4 * g + 2], acc[2 * u + 1][4 * g + 3]};
            a += eb2[2 * u][g];
            gt += eb2[2 * u + 1][g];
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = a[e] * wn_rcp(1.0f + wn_exp(-gt[e]));
            *reinterpret_cast<f32x4*>(wp2 + li * 272 + (u * 32 + 8 * g + 4 * hi) * 4) = o;
          }
#pragma unroll
        for (int it = 0; it < 8; ++it) {          // 32 rows x 16 pieces of 16 bytes
          const int q = it * 64 + lane, r = q >> 4, pc = q & 15;
          const f32x4 o = *reinterpret_cast<const f32x4*>(wp2 + r * 272 + pc * 16);
          if (m0 + r < p.M)
            *reinterpret_cast<f32x4*>(p.C + (int64_t)(m0 + r) * p.ldc + c2 / 2 + pc * 4) = o;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// 64-row blocks (round 4, second form).  With 32 rows per block every CU streams the WHOLE weight
// image from L2 for 32 rows of work: 1.6 MB per 512-column pass against 11.7 us of MFMA work --
// 64 B / clk / CU, the L1 fill rate, and 248 CUs x 64 B = the aggregate L2 rate; measured 24.4 us
// per pass (tools/bench_x6r.py, profiles/r06f_bench_x6r.txt), i.e. both limits at ~50 %.  Twice
// the rows per block halve the weight bytes per MFMA.  M = 7932 gives only 124 such blocks (no
// gain: half the CUs, twice the work each), M >= ~12 k fills the chip once -- config 3's 16231
// rows ran TWO rounds of 32-row blocks.
// 64 rows x 512 k as planes would be 192 KB: the rows stay fp32 in LDS (132 KB, row stride
// 2064 B: conflict-free lane = row reads) and a wave splits the two fragments of a k block in
// registers right before it multiplies them (per k block 16 values -> 48 bf16 against 48 MFMAs:
// the VALU work rides under the matrix pipe).  Wave w: columns [128 w, 128 w + 128) of a pass for
// BOTH 32-row tiles (8 accumulator tiles).  C / x_out / y / residual tiles cross the memory pipe
// as 128-byte row segments through a 4.5-KB wave patch.  The chained GEMM (EPI 3) needs no
// second image: the LayerNorm rows go back into the row area as fp32.
constexpr int WROWS = 64;
constexpr int RSTR = 2064;                    // bytes per fp32 row in LDS (516 floats)
constexpr int XROWS = WROWS * RSTR;           // 132,096 B
constexpr int QPST = 144;                     // quarter patch: 32 rows x (128 B + 16)
constexpr int QPATCH = 4 * 32 * QPST;         // 18,432 B

template <int EPI>
__global__ __launch_bounds__(256, 1) void x6r512w_kernel(X6RArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem5w[];
  __shared__ float red[2][4][WROWS];
  char* xrows = smem5w;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < 4);
  char* qp = smem5w + XROWS + wave * (32 * QPST);
  const int hi = lane >> 5, li = lane & 31;
  const int m0 = blockIdx.x * WROWS;
  const int Tn = (p.N + 31) >> 5;
  const int64_t kstride = (int64_t)Tn * X3_TILE;

  bf16x8 wf[PF5 + 1][NT5][3];
  auto w_base = [&](const void* W3, int ps) {
    return reinterpret_cast<const char*>(W3) + ((int64_t)(ps * 16 + wave * NT5) * 3) * X3_REC +
           lane * 16;
  };
  auto load_w = [&](const char* wb, int64_t kst, int ks) __attribute__((always_inline)) {
    const char* q = wb + ks * kst;
#pragma unroll
    for (int t = 0; t < NT5; ++t)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        wf[ks % (PF5 + 1)][t][pl] = *reinterpret_cast<const bf16x8*>(q + (t * 3 + pl) * X3_REC);
  };
  const char* wb = w_base(p.W3, 0);
#pragma unroll
  for (int s = 0; s < PF5; ++s) load_w(wb, kstride, s);

  // ---- the block's 64 rows -> LDS, fp32, whole rows per instruction (wave w: rows 16 w ..) ----
#pragma unroll
  for (int b8 = 0; b8 < 2; ++b8) {
    f32x4 rv[8][2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = min(m0 + wave * 16 + b8 * 8 + j, p.M - 1);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
        rv[j][hh] = *reinterpret_cast<const f32x4*>(p.A + (int64_t)r * p.lda + hh * 256 + lane * 4);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
        *reinterpret_cast<f32x4*>(xrows + (wave * 16 + b8 * 8 + j) * RSTR + hh * 1024 + lane * 16) =
            rv[j][hh];
  }
  __syncthreads();

  constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};
  f32x16 acc[2][NT5];
  // one 512-column pass over both row tiles; the fp32 fragments of k block ks + 1 are read from
  // LDS while the products of k block ks run, and split when their turn comes
  auto gemm_pass = [&](const char* wbp, int64_t kst) __attribute__((always_inline)) {   // (the first PF5 W loads are issued)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int t = 0; t < NT5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][t][r] = 0.f;
    f32x4 raw[2][2][2];                                  // [buffer][row tile][k quad]
    auto read_x = [&](int buf, int ks) __attribute__((always_inline)) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const char* q = xrows + (rt * 32 + li) * RSTR + ks * 64 + hi * 32;
        raw[buf][rt][0] = *reinterpret_cast<const f32x4*>(q);
        raw[buf][rt][1] = *reinterpret_cast<const f32x4*>(q + 16);
      }
    };
    // one quad (4 values) of row tile rt -> elements 4 h .. 4 h + 3 of the three planes
    auto split_quad = [&](bf16x8 (&X)[2][3], const f32x4& v, int rt, int h)
        __attribute__((always_inline)) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const Split3 sp = split3(v[e]);
        X[rt][0][4 * h + e] = sp.h0; X[rt][1][4 * h + e] = sp.h1; X[rt][2][4 * h + e] = sp.h2;
      }
    };
    // software pipeline: the products of k block ks run on fragments split one iteration
    // earlier; the four quads of k block ks + 1 are split BETWEEN the product groups of ks (one
    // wave per SIMD: VALU work in front of the MFMAs would leave the matrix pipe idle), and the
    // fp32 values of k block ks + 2 are on their way from LDS
    bf16x8 X[2][3], Xn[2][3];
    read_x(0, 0);
    read_x(1, 1);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      split_quad(X, raw[0][rt][0], rt, 0);
      split_quad(X, raw[0][rt][1], rt, 1);
    }
#pragma unroll
    for (int ks = 0; ks < KB5; ++ks) {
      if (ks + PF5 < KB5) load_w(wbp, kst, ks + PF5);
      f32x4 nx[2][2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) { nx[rt][0] = raw[(ks + 1) & 1][rt][0]; nx[rt][1] = raw[(ks + 1) & 1][rt][1]; }
      if (ks + 2 < KB5) read_x(ks & 1, ks + 2);
#pragma unroll
      for (int q = 0; q < 6; ++q) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int t = 0; t < NT5; ++t)
            acc[rt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks % (PF5 + 1)][t][PW[q]],
                                                                 X[rt][PX[q]], acc[rt][t], 0, 0, 0);
        if (ks + 1 < KB5 && q < 4) split_quad(Xn, nx[q >> 1][q & 1], q >> 1, q & 1);
      }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) X[rt][pl] = Xn[rt][pl];
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- tile traffic through the wave's quarter patch: lane = row <-> 128-byte row segments -----
  // v[g] = columns 8 g + 4 hi .. + 3 of the tile for row li
  auto tile_out = [&](const f32x4 (&v)[4], float* base, int ld, int row0, int col)
      __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<f32x4*>(qp + li * QPST + (8 * g + 4 * hi) * 4) = v[g];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int q = it * 64 + lane, r = q >> 3, pc = q & 7;
      const f32x4 o = *reinterpret_cast<const f32x4*>(qp + r * QPST + pc * 16);
      if (row0 + r < p.M)
        *reinterpret_cast<f32x4*>(base + (int64_t)(row0 + r) * ld + col + pc * 4) = o;
    }
  };
  auto tile_in = [&](f32x4 (&v)[4], const float* base, int ld, int row0, int col)
      __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int q = it * 64 + lane, r = q >> 3, pc = q & 7;
      const int rc = min(row0 + r, p.M - 1);
      *reinterpret_cast<f32x4*>(qp + r * QPST + pc * 16) =
          *reinterpret_cast<const f32x4*>(base + (int64_t)rc * ld + col + pc * 4);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
      v[g] = *reinterpret_cast<const f32x4*>(qp + li * QPST + (8 * g + 4 * hi) * 4);
  };
  constexpr int WCOLS = NT5 * 32;

  if constexpr (EPI == 0) {
    const int npass = p.N / 512;
    for (int ps = 0; ps < npass; ++ps) {
      f32x4 ebw[NT5][4];     // (one batch under the GEMM, see x6r512_kernel)
#pragma unroll
      for (int t = 0; t < NT5; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          ebw[t][g] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (p.bias) ebw[t][g] = *reinterpret_cast<const f32x4*>(p.bias + ps * 512 + wave * WCOLS + t * 32 + 8 * g + 4 * hi);
        }
      const char* wbp = w_base(p.W3, ps);
      gemm_pass(wbp, kstride);
      // the next pass's first W fragments go out BEFORE this pass's stores: vmcnt retires in
      // order, so loads queued behind 128 KB of stores would hold the next pass's first MFMAs
      // until the store burst of all blocks has drained
      if (ps + 1 < npass) {
        const char* wbn = w_base(p.W3, ps + 1);
#pragma unroll
        for (int s = 0; s < PF5; ++s) load_w(wbn, kstride, s);
      }
      const int col0 = ps * 512 + wave * WCOLS;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int t = 0; t < NT5; ++t) {
          f32x4 v[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            v[g] = f32x4{acc[rt][t][4 * g], acc[rt][t][4 * g + 1], acc[rt][t][4 * g + 2],
                         acc[rt][t][4 * g + 3]};
            v[g] += ebw[t][g];
          }
          tile_out(v, p.C, p.ldc, m0 + rt * 32, col0 + t * 32);
        }
    }
  } else {
    f32x4 ebw[NT5][4];     // (one batch under the GEMM, see x6r512_kernel)
#pragma unroll
    for (int t = 0; t < NT5; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        ebw[t][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) ebw[t][g] = *reinterpret_cast<const f32x4*>(p.bias + wave * WCOLS + t * 32 + 8 * g + 4 * hi);
      }
    gemm_pass(wb, kstride);
    const int col0 = wave * WCOLS;
    f32x4 v[2][NT5][4];
    float s1[2] = {0.f, 0.f};
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int t = 0; t < NT5; ++t) {
        f32x4 rs[4];
        tile_in(rs, p.resid, p.ldr, m0 + rt * 32, col0 + t * 32);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = col0 + t * 32 + 8 * g + 4 * hi;
          f32x4 a = f32x4{acc[rt][t][4 * g], acc[rt][t][4 * g + 1], acc[rt][t][4 * g + 2],
                          acc[rt][t][4 * g + 3]};
          a += ebw[t][g];
          v[rt][t][g] = rs[g] + p.alpha * a;
          s1[rt] += (v[rt][t][g][0] + v[rt][t][g][1]) + (v[rt][t][g][2] + v[rt][t][g][3]);
        }
        tile_out(v[rt][t], p.x_out, p.ldx, m0 + rt * 32, col0 + t * 32);
      }
    // LayerNorm statistics of the 64 rows: lane pair, then the four waves (mean, then the centred
    // squares)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      s1[rt] += __shfl_xor(s1[rt], 32, 64);
      if (hi == 0) red[0][wave][rt * 32 + li] = s1[rt];
    }
    __syncthreads();                            // (also: every wave is done with the row area)
    float mean[2], rstd[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int r = rt * 32 + li;
      mean[rt] = ((red[0][0][r] + red[0][1][r]) + (red[0][2][r] + red[0][3][r])) * (1.0f / K5);
      float s2 = 0.f;
#pragma unroll
      for (int t = 0; t < NT5; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = v[rt][t][g][e] - mean[rt];
            s2 += d * d;
          }
      s2 += __shfl_xor(s2, 32, 64);
      if (hi == 0) red[1][wave][r] = s2;
    }
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int r = rt * 32 + li;
      const float var = ((red[1][0][r] + red[1][1][r]) + (red[1][2][r] + red[1][3][r])) * (1.0f / K5);
      rstd[rt] = 1.0f / sqrtf(var + p.eps);
    }
#pragma unroll
    for (int t = 0; t < NT5; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = col0 + t * 32 + 8 * g + 4 * hi;
        const f32x4 w = *reinterpret_cast<const f32x4*>(p.ln_w + c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.ln_b + c);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[rt][t][g][e] = (v[rt][t][g][e] - mean[rt]) * rstd[rt] * w[e] + b[e];
      }
    if (p.y != nullptr) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int t = 0; t < NT5; ++t) tile_out(v[rt][t], p.y, p.ldy, m0 + rt * 32, col0 + t * 32);
    }
    if constexpr (EPI == 1) {
      if (p.y3 != nullptr) {      // y as its X3 plane image (x6.h x3_store_tile)
        const int tiles_m = (p.M + 31) >> 5;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
          if ((m0 >> 5) + rt < tiles_m) {
#pragma unroll
            for (int t = 0; t < NT5; ++t)
              x3_store_tile(v[rt][t], p.y3, (col0 + t * 32) >> 4, tiles_m, (m0 >> 5) + rt, hi, li);
          }
      }
    }
    if constexpr (EPI == 3) {
      // ---- chained: C = GLU(y W3b^T + bias2), N2 = 1024: the LayerNorm rows go back into the
      // row area (every wave left it at the barriers above) and the second GEMM reads them like
      // the first
      const char* wb2 = w_base(p.W3b, 0);
      const int64_t kst2 = (int64_t)32 * X3_TILE;
#pragma unroll
      for (int s = 0; s < PF5; ++s) load_w(wb2, kst2, s);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int t = 0; t < NT5; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(xrows + (rt * 32 + li) * RSTR +
                                      (col0 + t * 32 + 8 * g + 4 * hi) * 4) = v[rt][t][g];
      __syncthreads();
      for (int ps = 0; ps < 2; ++ps) {
        f32x4 ebw2[NT5][4];     // (one batch under the GEMM, see x6r512_kernel)
#pragma unroll
        for (int t = 0; t < NT5; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            ebw2[t][g] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias2) ebw2[t][g] = *reinterpret_cast<const f32x4*>(p.bias2 + ps * 512 + wave * WCOLS + t * 32 + 8 * g + 4 * hi);
          }
        const char* wbp = w_base(p.W3b, ps);
        gemm_pass(wbp, kst2);
        if (ps == 0) {                            // (ahead of the stores, see EPI 0)
          const char* wbn = w_base(p.W3b, 1);
#pragma unroll
          for (int s = 0; s < PF5; ++s) load_w(wbn, kst2, s);
        }
        const int c2 = ps * 512 + wave * WCOLS;   // first column of the wave's tiles in the image
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            f32x4 o[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int c = c2 + 2 * u * 32 + 8 * g + 4 * hi;
              f32x4 a = f32x4{acc[rt][2 * u][4 * g], acc[rt][2 * u][4 * g + 1],
                              acc[rt][2 * u][4 * g + 2], acc[rt][2 * u][4 * g + 3]};
              f32x4 gt = f32x4{acc[rt][2 * u + 1][4 * g], acc[rt][2 * u + 1][4 * g + 1],
                               acc[rt][2 * u + 1][4 * g + 2], acc[rt][2 * u + 1][4 * g + 3]};
              a += ebw2[2 * u][g];
              gt += ebw2[2 * u + 1][g];
#pragma unroll
              for (int e = 0; e < 4; ++e)
                o[g][e] = a[e] * wn_rcp(1.0f + wn_exp(-gt[e]));
            }
            tile_out(o, p.C, p.ldc, m0 + rt * 32, c2 / 2 + u * 32);
          }
      }
    }
  }
}

template <int EPI>
int launch_x6r512w(const X6RArgs& a, hipStream_t s) {
  const size_t lds = (size_t)XROWS + QPATCH;
  auto kern = x6r512w_kernel<EPI>;
  WN_MAX_DYN_LDS(kern, lds);
  hipLaunchKernelGGL(kern, dim3(cdiv(a.M, WROWS)), dim3(256), lds, s, a);
  WN_HIP(hipGetLastError());
  return 0;
}

template <int EPI>
int launch_x6r512(const X6RArgs& a, hipStream_t s) {
  const size_t lds = (size_t)XIMG + HPATCH;
  auto kern = x6r512_kernel<EPI>;
  WN_MAX_DYN_LDS(kern, lds);
  hipLaunchKernelGGL(kern, dim3(cdiv(a.M, 32)), dim3(256), lds, s, a);
  WN_HIP(hipGetLastError());
  return 0;
}

}  // namespace

bool gemm_x6r512_supported(int M, int N, int epi) {
  if (M <= 0) return false;
  if (epi == 1 || epi == 3) return N == 512;
  if (epi == 0) return N % 512 == 0 && N >= 512 && N <= 2048;
  return false;                                 // (GLU alone: the chain covers it)
}


// 64-row blocks once they cover at least three quarters of the 256 CUs (M >= 12288)
static bool x6r512_wide(const X6RArgs& a) {
  if (tune().x6r512_rows == 64) return true;
  if (tune().x6r512_rows == 32) return false;
  return cdiv(a.M, WROWS) >= 192;
}

int gemm_x6r512(const X6RArgs& a, hipStream_t s) {
  WN_CHECK(!a.pro_P, "gemm_x6r512: no prologue fold at K = 512 (measured slower than ffn_reduce_ln)");
  WN_CHECK(a.A && a.W3 && a.lda % 4 == 0 && gemm_x6r512_supported(a.M, a.N, a.epi),
           "gemm_x6r512: shape");
  const bool wide = x6r512_wide(a);
  if (a.epi == 1) {
    WN_CHECK(a.resid && a.x_out && a.ln_w && a.ln_b && (a.y || a.y3) && a.ldr % 4 == 0 && a.ldx % 4 == 0 &&
                 a.ldy % 4 == 0, "gemm_x6r512: row-LN epilogue arguments");
    return wide ? launch_x6r512w<1>(a, s) : launch_x6r512<1>(a, s);
  }
  if (a.epi == 3) {
    WN_CHECK(a.resid && a.x_out && a.ln_w && a.ln_b && a.ldr % 4 == 0 && a.ldx % 4 == 0 &&
                 (a.y == nullptr || a.ldy % 4 == 0) && a.W3b && a.C && a.ldc % 4 == 0,
             "gemm_x6r512: chained row-LN + GLU arguments");
    return wide ? launch_x6r512w<3>(a, s) : launch_x6r512<3>(a, s);
  }
  WN_CHECK(a.C && a.ldc % 4 == 0, "gemm_x6r512: no output");
  return wide ? launch_x6r512w<0>(a, s) : launch_x6r512<0>(a, s);
}

}  // namespace wn
