// fp32 GEMM whose block owns COMPLETE output rows (N = d_model = 256), fused with the
// residual add and the LayerNorm that follows it:
//     x <- resid + alpha (A W^T + bias)          (attention out-projection / pointwise_conv2,
//     y <- LayerNorm(x; ln_w, ln_b, eps)          attention.py:176, convolution.py:148;
//                                                 encoder_layer.py:238-240, 251-253)
// Before: a 64x64-tile GEMM at 0.37 of the fp32-MFMA rate (496 small blocks, K = 256:
// prologue / epilogue dominated, profiles/r02a) followed by a separate LayerNorm launch --
// 24 + 24 launches per decode step.  Here 32 rows x 256 columns per block (248 blocks at
// M = 7932: the 256 CUs once), 8 waves side by side along N (wave tile 32 x 32), W and the
// block's A rows streamed through the DMA ring of ffn_fused.hip (stage = 32 A rows + 256 W
// rows x 32 k, mid-stage barrier, counted vmcnt), and the row statistics reduced across
// the 8 waves through LDS in the epilogue (two-pass mean / variance like layernorm_kernel).
// Exact fp32 (v_mfma_f32_32x32x2_f32).
#include "common.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace wn {

namespace {

constexpr int RBM = 32;                       // rows per block
constexpr int RN = 256;                       // columns = d_model
constexpr int RSTG = (RBM + RN) * 128;        // 36 KB per stage
constexpr int RRING = 4;

__global__ __launch_bounds__(512) void gemm_rowln_kernel(RowLnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_r[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  float* red = reinterpret_cast<float*>(smem_r + RRING * RSTG);   // [8 waves][32 rows] + [32]

  const int m0 = blockIdx.x * RBM;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, li = lane & 31;
  const int nk = p.K / 32;

  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.A), 0, (int)min((int64_t)p.M * p.lda * 4, (int64_t)0x7fffffff),
      0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.W), 0, (int)min((int64_t)RN * p.K * 4, (int64_t)0x7fffffff),
      0x00020000);
  // stage rows 0..31: A; 32..287: W.  Piece q of a wave covers 8 rows; lane -> row +
  // (lane >> 3), LDS slot lane & 7, source slot = slot ^ ((row >> 1) & 7).  Every wave
  // issues 4 W pieces and 1 A piece (waves 4-7 repeat the A pieces of waves 0-3: the same
  // bytes to the same place, which keeps the DMA count per wave and stage uniform).
  unsigned vA, vW[4];
  {
    const int rr = lane >> 3;
    const int ar = (wave & 3) * 8 + rr;                        // A row 0..31
    vA = (unsigned)min(m0 + ar, p.M - 1) * (unsigned)p.lda * 4u +
         (unsigned)(((lane & 7) ^ ((ar >> 1) & 7)) * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int wr = (q * 8 + wave) * 8 + rr;                  // W row 0..255
      // stage row = 32 + wr: (32 + wr) >> 1 & 7 == (wr >> 1) & 7
      vW[q] = (unsigned)wr * (unsigned)p.K * 4u + (unsigned)(((lane & 7) ^ ((wr >> 1) & 7)) * 16);
    }
  }
  auto issue = [&](int g) {
    g = min(g, nk - 1);
    char* dst = smem_r + (g % RRING) * RSTG;
    const int koff = g * 128;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(dst + (wave & 3) * 1024), 16, vA, koff,
                                             0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rw, (lds_ptr)(dst + RBM * 128 + (q * 8 + wave) * 1024), 16, vW[q], koff, 0, 0);
  };
  const int sw = (lane >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) foff[kk] = li * 128 + (((kk * 2 + hi) ^ sw) << 4);

  // epilogue operands first: no ordinary load may be pending once the DMA ring runs
  const int col = wave * 32 + li;
  const float bias = p.bias ? p.bias[col] : 0.f;
  const float lnw = p.ln_w[col], lnb = p.ln_b[col];
  float rs[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    rs[r] = (p.resid && row < p.M) ? p.resid[(int64_t)row * p.ldr + col] : 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(rs[r]));

  struct Frag { f32x4 a, b; };
  auto load = [&](int g, int kk) {
    const char* st = smem_r + (g % RRING) * RSTG;
    Frag f;
    f.a = *reinterpret_cast<const f32x4*>(st + foff[kk]);
    f.b = *reinterpret_cast<const f32x4*>(st + (RBM + wave * 32) * 128 + foff[kk]);
    return f;
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  auto mma = [&](const Frag& f) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[s], f.b[s], acc, 0, 0, 0);
  };

  issue(0); issue(1); issue(2);
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");     // stage 0 of this wave landed
  __builtin_amdgcn_s_barrier();
  Frag x0 = load(0, 0), x1 = load(0, 1);
  for (int g = 0; g < nk; ++g) {
    const Frag y0 = load(g, 2), y1 = load(g, 3);
    mma(x0); mma(x1);
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");    // stage g+1 landed (g+2 in flight)
    __builtin_amdgcn_s_barrier();
    issue(g + 3);
    if (g + 1 < nk) { x0 = load(g + 1, 0); x1 = load(g + 1, 1); }
    mma(y0); mma(y1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // trailing (clamped) DMA

  // ---- x = resid + alpha (acc + bias); row statistics across the 8 waves ----------------
  float v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = rs[r] + p.alpha * (acc[r] + bias);
  float* part = red;               // [8][32]
  float* stat = red + 8 * 32;      // [32]
  auto row_reduce = [&](float (&t)[16]) {
    // sum over the 32 lanes of this half (same hi): xor 1 .. 16 stay inside the half
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float s = t[r];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      t[r] = s;
    }
    if (li == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) part[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi] = t[r];
    }
    __syncthreads();
    if (tid < 32) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += part[w * 32 + tid];
      stat[tid] = s * (1.0f / RN);
    }
    __syncthreads();
  };
  float t[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) t[r] = v[r];
  row_reduce(t);
  float mean[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) mean[r] = stat[(r & 3) + 8 * (r >> 2) + 4 * hi];
  __syncthreads();                 // stat is rewritten below
#pragma unroll
  for (int r = 0; r < 16; ++r) { const float dlt = v[r] - mean[r]; t[r] = dlt * dlt; }
  row_reduce(t);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;
    const int row = m0 + lr;
    if (row < p.M) {
      const float rstd = 1.0f / sqrtf(stat[lr] + p.eps);
      p.x_out[(int64_t)row * p.ldx + col] = v[r];
      p.y[(int64_t)row * p.ldy + col] = (v[r] - mean[r]) * rstd * lnw + lnb;
    }
  }
}

}  // namespace


bool gemm_rowln_supported(int M, int N, int K) {
  return tune().gemm_rowln != 0 && N == RN && K % 32 == 0 && K >= 96 && M > 0 &&
         (int64_t)M * K * 4 < (int64_t(1) << 31);
}

int gemm_rowln(const RowLnArgs& a, hipStream_t s) {
  WN_CHECK(a.A && a.W && a.x_out && a.y && a.ln_w && a.ln_b, "gemm_rowln: null argument");
  WN_CHECK(a.N == RN && a.K % 32 == 0 && a.K >= 96 && a.lda % 4 == 0, "gemm_rowln: shape");
  const size_t lds = (size_t)RRING * RSTG + (8 * 32 + 32) * sizeof(float);
  WN_MAX_DYN_LDS(gemm_rowln_kernel, lds);
  hipLaunchKernelGGL(gemm_rowln_kernel, dim3(cdiv(a.M, RBM)), dim3(512), lds, s, a);
  WN_HIP(hipGetLastError());
  return 0;
}

}  // namespace wn
