// Epilogue shared by the GEMM kernels (gemm.hip: fp32 operands, gemm_bf16.hip:
// bf16 operands): bias, SiLU / ReLU / exact GELU, alpha, residual add, GLU, on
// the fp32 accumulators of WGM x WGN waves with MT x NT 32x32 MFMA tiles each.
// The C/D layout of the 32x32 MFMA does not depend on the operand type:
//   col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
#pragma once
#include "common.h"

namespace wn {
namespace {

__device__ __forceinline__ float silu_fast(float x) {
  // x * sigmoid(x) on v_exp_f32 / v_rcp_f32 (each ~1 ulp); __frcp_rn would be
  // the correctly rounded division sequence (v_div_scale / fmas / fixup)
  return x * wn_rcp(1.0f + wn_exp(-x));
}

// CH: C is a bf16 matrix (p.C reinterpreted, p.ldc in bf16 elements) -- the
// bf16-storage mode's FFN hidden tensor; only without residual / GLU.
template <int BM, int BN, int WGM, int WGN, int ACT, bool RESID, bool GLU,
          bool CH = false>
__device__ __forceinline__ void gemm_epilogue(
    const GemmArgs& p, f32x16 (&acc)[BM / WGM / 32][BN / WGN / 32], int m0,
    int n0, int wm, int wn_, int lane) {
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  static_assert(!CH || (!RESID && !GLU), "bf16 C: plain / activation epilogues only");
  const int col_in = lane & 31;
  const int row_hi = (lane >> 5) * 4;
  if constexpr (GLU) {
    const int cbase = n0 + wn_ * WTN;  // permuted column of the 'a' half
    const int ca = cbase + col_in, cg = cbase + 32 + col_in;
    const int cout = cbase / 2 + col_in;
    const bool cok = cg < p.N;
    const float ba = (p.bias && cok) ? p.bias[ca] : 0.0f;
    const float bg = (p.bias && cok) ? p.bias[cg] : 0.0f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int row0 = m0 + wm * WTM + i * 32 + row_hi;
      float* cp = p.C + (int64_t)row0 * p.ldc + cout;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float a = acc[i][0][r] + ba;
        const float g = acc[i][1][r] + bg;
        v[r] = a * wn_rcp(1.0f + wn_exp(-g));
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        if (row0 + dr < p.M && cok) cp[dr * p.ldc] = v[r];
      }
    }
    return;
  } else {
    // Values first (no branches, so the 16 results of a tile overlap their
    // exp / rcp latencies), then the stores; only the ragged last tiles pay for
    // per-row predicates.
    const bool full = (m0 + BM <= p.M) && (n0 + BN <= p.N);  // block-uniform
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = n0 + wn_ * WTN + j * 32 + col_in;
      const bool cok = col < p.N;
      const float b = (p.bias && cok) ? p.bias[col] : 0.0f;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row0 = m0 + wm * WTM + i * 32 + row_hi;
        float* cp = p.C + (int64_t)row0 * p.ldc + col;
        const float* rp = RESID ? p.resid + (int64_t)row0 * p.ldr + col : nullptr;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float x = acc[i][j][r] + b;
          if (ACT == ACT_SILU) x = silu_fast(x);
          if (ACT == ACT_RELU) x = fmaxf(x, 0.0f);
          if (ACT == ACT_GELU) x = 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
          v[r] = x * p.alpha;
        }
        if constexpr (CH) {
          __bf16* hp = reinterpret_cast<__bf16*>(p.C) + (int64_t)row0 * p.ldc + col;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            if (full || (row0 + dr < p.M && cok)) hp[dr * p.ldc] = (__bf16)v[r];
          }
        } else if (full) {
          if (RESID) {
            float rr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) rr[r] = rp[((r & 3) + 8 * (r >> 2)) * p.ldr];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += rr[r];
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) cp[((r & 3) + 8 * (r >> 2)) * p.ldc] = v[r];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            if (row0 + dr < p.M && cok) {
              float x = v[r];
              if (RESID) x += rp[dr * p.ldr];
              cp[dr * p.ldc] = x;
            }
          }
        }
      }
    }
  }
}

// XCD-aware tile assignment (bijective for any grid size): the N tiles of one
// M panel run on the same XCD so the A panel is fetched into one L2, not eight.
__device__ __forceinline__ int xcd_block_order(int bid, int nblk) {
  const int q = nblk / 8, r = nblk % 8;
  const int xcd = bid % 8, slot = bid / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

}  // namespace
}  // namespace wn
