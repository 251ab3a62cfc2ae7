// Row-block fp32 GEMM on the bf16 matrix cores (six plane products, gemm_x6.hip) for the
// encoder's SMALL projections at d_model = 256 -- K = 256, N = 256 .. 768, M = a few thousand
// rows: the attention output projection and pointwise_conv2 fused with their residual add and
// the LayerNorm that follows (attention.py:176, convolution.py:148; encoder_layer.py:238-240,
// 251-253), and the QKV projection (attention.py:109-131).
//
// Why (round 3): these GEMMs are 1-3 GFLOP each -- 7-20 us of v_mfma_f32 work that took
// 23-37 us per launch (MFMA busy 0.23-0.48, profiles/r03i): tile GEMMs at this size are all
// prologue and epilogue, and the six-product tile GEMM needs a separate pass that writes the
// plane image of A.  Here the structure of ffn_x6f.hip is reused instead:
//   * a block = 4 waves = ONE wave per SIMD owns 32 rows of A; every wave loads those rows as
//     fp32 (lane = row, its k half) and splits them into the three bf16 planes in registers --
//     the "B" operand fragments of all 16 k blocks, 192 registers, no plane image of A;
//   * the waves split N: wave w computes columns [w N / 4, (w + 1) N / 4) = NT tiles of 32, so
//     each W fragment is read by exactly one wave -- straight from the weight plane image in
//     L2 into registers (16 B per lane, one 1-KB record per instruction, PF k blocks ahead):
//     no LDS, no DMA ring, no barrier in the main loop; 248 blocks re-read the same 0.4-1.2 MB
//     image, which stays in every XCD's L2;
//   * W fragment = the instruction's "A" operand, so a lane ends up with ONE row and the columns
//     8 g + 4 (lane / 32) + q of every tile: bias / residual / stores are 16-byte pieces of a row,
//     and a row's LayerNorm statistics are 32 NT values per lane, one exchange with lane ^ 32 and
//     one LDS round between the four waves (two passes: mean, then the centred squares, like
//     layernorm_kernel).
// Grid = ceil(M / 32) blocks (248 at M = 7932: the 256 CUs once).
#include "common.h"
#include "kernels.h"
#include "x6.h"

namespace wn {

namespace {

constexpr int RK = 256;              // K of this kernel
constexpr int RKB = RK / 16;         // 16 k blocks

// EPI 0: C = acc + bias; EPI 1: x_out = resid + alpha (acc + bias), y = LayerNorm(x_out)
template <int NT, int EPI, int PF>
__global__ __launch_bounds__(256, 1) void x6r_kernel(X6RArgs p) {
  __shared__ float red[2][4][32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < 4);
  const int hi = lane >> 5, li = lane & 31;
  const int m0 = blockIdx.x * 32;
  const int row = m0 + li;
  const int rowc = min(row, p.M - 1);

  // ---- A rows: fp32, 8 consecutive floats per k block and lane ---------------------------------
  f32x4 xa[RKB], xb[RKB];
  {
    const float* ar = p.A + (int64_t)rowc * p.lda + hi * 8;
#pragma unroll
    for (int ks = 0; ks < RKB; ++ks) {
      xa[ks] = *reinterpret_cast<const f32x4*>(ar + ks * 16);
      xb[ks] = *reinterpret_cast<const f32x4*>(ar + ks * 16 + 4);
    }
  }
  // ---- W fragments: records [k block][tile][plane] of the weight image, this wave's NT tiles ---
  const int Tn = (p.N + 31) >> 5;
  const char* wb = reinterpret_cast<const char*>(p.W3) + ((int64_t)wave * NT * 3) * X3_REC +
                   lane * 16;
  const int64_t kstride = (int64_t)Tn * X3_TILE;
  bf16x8 wf[PF + 1][NT][3];
  auto load_w = [&](int ks) {
    const char* q = wb + ks * kstride;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        wf[ks % (PF + 1)][t][pl] = *reinterpret_cast<const bf16x8*>(q + (t * 3 + pl) * X3_REC);
  };
#pragma unroll
  for (int s = 0; s < PF; ++s) load_w(s);

  // exact three-way bf16 split of the rows in registers (x6.h)
  bf16x8 X[RKB][3];
#pragma unroll
  for (int ks = 0; ks < RKB; ++ks) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const Split3 sa = split3(xa[ks][e]), sb = split3(xb[ks][e]);
      X[ks][0][e] = sa.h0; X[ks][1][e] = sa.h1; X[ks][2][e] = sa.h2;
      X[ks][0][4 + e] = sb.h0; X[ks][1][4 + e] = sb.h1; X[ks][2][4 + e] = sb.h2;
    }
  }

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // plane products, the small ones first: (W plane, activation plane); the tiles alternate so
  // that no MFMA waits for its predecessor's accumulator
  constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
  for (int ks = 0; ks < RKB; ++ks) {
    if (ks + PF < RKB) load_w(ks + PF);
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks % (PF + 1)][t][PW[q]],
                                                         X[ks][PX[q]], acc[t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue: lane = row, registers = columns col0 + 32 t + 8 g + 4 hi + e ------------------
  const int col0 = wave * NT * 32;
  if constexpr (EPI == 0) {
    if (row < p.M) {
      float* crow = p.C + (int64_t)row * p.ldc;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = col0 + t * 32 + 8 * g + 4 * hi;
          if (c >= p.N) continue;
          f32x4 v = f32x4{acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
          if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + c);
          *reinterpret_cast<f32x4*>(crow + c) = v;
        }
    }
  } else {
    f32x4 v[NT][4];
    float s1 = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = col0 + t * 32 + 8 * g + 4 * hi;
        f32x4 a = f32x4{acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
        if (p.bias) a += *reinterpret_cast<const f32x4*>(p.bias + c);
        const f32x4 r = *reinterpret_cast<const f32x4*>(p.resid + (int64_t)rowc * p.ldr + c);
        v[t][g] = r + p.alpha * a;
        s1 += (v[t][g][0] + v[t][g][1]) + (v[t][g][2] + v[t][g][3]);
        if (row < p.M) *reinterpret_cast<f32x4*>(p.x_out + (int64_t)row * p.ldx + c) = v[t][g];
      }
    // mean over the row's N columns: lane pair, then the four waves
    s1 += __shfl_xor(s1, 32, 64);
    if (hi == 0) red[0][wave][li] = s1;
    __syncthreads();
    const float mean =
        ((red[0][0][li] + red[0][1][li]) + (red[0][2][li] + red[0][3][li])) / (float)p.N;
    float s2 = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
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
        ((red[1][0][li] + red[1][1][li]) + (red[1][2][li] + red[1][3][li])) / (float)p.N;
    const float rstd = 1.0f / sqrtf(var + p.eps);
    if (row < p.M) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = col0 + t * 32 + 8 * g + 4 * hi;
          const f32x4 w = *reinterpret_cast<const f32x4*>(p.ln_w + c);
          const f32x4 b = *reinterpret_cast<const f32x4*>(p.ln_b + c);
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (v[t][g][e] - mean) * rstd * w[e] + b[e];
          *reinterpret_cast<f32x4*>(p.y + (int64_t)row * p.ldy + c) = o;
        }
    }
  }
}

template <int NT, int EPI, int PF>
int launch_x6r(const X6RArgs& a, hipStream_t s) {
  hipLaunchKernelGGL((x6r_kernel<NT, EPI, PF>), dim3(cdiv(a.M, 32)), dim3(256), 0, s, a);
  WN_HIP(hipGetLastError());
  return 0;
}

}  // namespace

int g_x6r = 1;       // wn_tune_set("x6r"): 0 = the v_mfma_f32 row-LN GEMM / tile GEMMs (A/B, tests)

bool gemm_x6r_supported(int M, int N, int K, int epi) {
  if (K != RK || M <= 0) return false;
  if (epi == 1) return N == 256;
  return N == 256 || N == 512 || N == 768;
}

int gemm_x6r(const X6RArgs& a, hipStream_t s) {
  WN_CHECK(a.A && a.W3 && a.lda % 4 == 0 && gemm_x6r_supported(a.M, a.N, RK, a.epi),
           "gemm_x6r: shape");
  if (a.epi == 1) {
    WN_CHECK(a.resid && a.x_out && a.ln_w && a.ln_b && a.y && a.ldr % 4 == 0 && a.ldx % 4 == 0 &&
                 a.ldy % 4 == 0, "gemm_x6r: row-LN epilogue arguments");
    return launch_x6r<2, 1, 3>(a, s);
  }
  WN_CHECK(a.C && a.ldc % 4 == 0, "gemm_x6r: no output");
  switch (a.N) {
    case 256: return launch_x6r<2, 0, 3>(a, s);
    case 512: return launch_x6r<4, 0, 2>(a, s);
    default: return launch_x6r<6, 0, 1>(a, s);
  }
}

}  // namespace wn
