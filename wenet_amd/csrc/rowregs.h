// One wave per row with the row in registers, and the LayerNorm over it: shared by the
// row kernels of encoder_kernels.hip and the prologues that do their work inside a GEMM
// (gemm_x6r.hip DWC).  A private copy per translation unit (-fno-gpu-rdc).
#pragma once
#include "common.h"
#include "kernels.h"

namespace wn {
namespace {

// ===========================================================================
// LayerNorm -- torch.nn.LayerNorm(d, eps) as used by encoder_layer.py:166-184.
// One wave per row; the row lives in registers (E = D/64 values per lane),
// two-pass mean / variance, wave-shuffle reductions, 16-byte accesses.
template <int E>
struct RowRegs {
  // widest vector that divides the per-lane element count
  static constexpr int VEC = E % 4 == 0 ? 4 : (E % 2 == 0 ? 2 : 1);
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  float v[E];
  __device__ __forceinline__ void load(const float* p, int lane) {
#pragma unroll
    for (int j = 0; j < E / VEC; ++j) {
      if constexpr (VEC == 1) {
        v[j] = p[j * 64 + lane];
      } else {
        const vec_t t = *reinterpret_cast<const vec_t*>(p + j * 64 * VEC + lane * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[j * VEC + e] = t[e];
      }
    }
  }
  __device__ __forceinline__ void store(float* p, int lane) const {
#pragma unroll
    for (int j = 0; j < E / VEC; ++j) {
      if constexpr (VEC == 1) {
        p[j * 64 + lane] = v[j];
      } else {
        vec_t t;
#pragma unroll
        for (int e = 0; e < VEC; ++e) t[e] = v[j * VEC + e];
        *reinterpret_cast<vec_t*>(p + j * 64 * VEC + lane * VEC) = t;
      }
    }
  }
};

template <int E>
__device__ __forceinline__ void ln_inplace(RowRegs<E>& r, const float* w,
                                           const float* b, int lane,
                                           float eps) {
  constexpr int D = E * 64;
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) s += r.v[e];
  const float mean = wave_sum(s) * (1.0f / D);
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const float d = r.v[e] - mean;
    q += d * d;
  }
  const float var = wave_sum(q) * (1.0f / D);
  const float rstd = 1.0f / sqrtf(var + eps);
  RowRegs<E> ww, bb;
  ww.load(w, lane);
  bb.load(b, lane);
#pragma unroll
  for (int e = 0; e < E; ++e)
    r.v[e] = (r.v[e] - mean) * rstd * ww.v[e] + bb.v[e];
}

// The middle of the convolution module for R = 8 consecutive packed rows row0 .. row0 + 7 by one
// wave (convolution.py:119-146): depthwise conv over time + LayerNorm / eval-BatchNorm affine +
// SiLU of the GLU output a.x -- dwconv_tiled_kernel's operations per output row in its order
// (bias, the taps in ascending order with the pad rule evaluated against the output row's own
// utterance, norm, SiLU), as the prologue of the row-block GEMMs (gemm_x6r.hip, gemm_x6r512.hip).
// A group of 8 taps shares its 15 window rows.  Rows outside the batch / an utterance come back
// as zeros with on[r] = false.  (Needs kernels.h for DwConvArgs.)
template <int E>
__device__ __forceinline__ void dwconv_rows8(const DwConvArgs& a, int row0, int lane,
                                             RowRegs<E> (&acc)[8], bool (&on)[8]) {
  constexpr int R = 8, TG = 8, NWIN = R + TG - 1;
  const int lpad = a.causal ? a.K - 1 : (a.K - 1) / 2;
  int u_l = -1, off_l = 0, len_l = 0;
  if (lane < R && row0 + lane < a.M) u_l = a.row_utt[row0 + lane];
  if (u_l >= 0) {
    off_l = a.off[u_l];
    len_l = a.len[u_l];
  }
  int t_r[R], len_r[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int u = __builtin_amdgcn_readlane(u_l, r);
    t_r[r] = row0 + r - __builtin_amdgcn_readlane(off_l, r);
    len_r[r] = __builtin_amdgcn_readlane(len_l, r);
    on[r] = u >= 0 && t_r[r] < len_r[r];
  }
  RowRegs<E> cp;
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r].load(a.bias, lane);
  cp.load(a.cpad, lane);
  for (int k0 = 0; k0 < a.K; k0 += TG) {
    RowRegs<E> wk[TG], xw[NWIN];
#pragma unroll
    for (int i = 0; i < NWIN; ++i) {
      const int q = min(max(row0 + k0 - lpad + i, 0), a.M - 1);
      xw[i].load(a.x + (int64_t)q * a.ldx, lane);
    }
#pragma unroll
    for (int i = 0; i < TG; ++i)
      if (k0 + i < a.K) wk[i].load(a.wt + (int64_t)(k0 + i) * (E * 64), lane);
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int j = 0; j < TG; ++j) {
        const int k = k0 + j, tt = t_r[r] + k - lpad;
        if (on[r] && k < a.K) {
          if (tt >= 0 && tt < len_r[r]) {
#pragma unroll
            for (int e = 0; e < E; ++e) acc[r].v[e] = fmaf(wk[j].v[e], xw[r + j].v[e], acc[r].v[e]);
          } else if ((tt < 0 && a.causal) || (tt >= len_r[r] && tt < a.t_max)) {
#pragma unroll
            for (int e = 0; e < E; ++e) acc[r].v[e] = fmaf(wk[j].v[e], cp.v[e], acc[r].v[e]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (on[r]) {
      if (a.norm_mode == 0) {
        ln_inplace<E>(acc[r], a.ln_w, a.ln_b, lane, a.eps);
      } else {
        RowRegs<E> sc, sh;
        sc.load(a.ln_w, lane);
        sh.load(a.ln_b, lane);
#pragma unroll
        for (int e = 0; e < E; ++e) acc[r].v[e] = fmaf(acc[r].v[e], sc.v[e], sh.v[e]);
      }
#pragma unroll
      for (int e = 0; e < E; ++e) acc[r].v[e] = silu_f(acc[r].v[e]);
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) acc[r].v[e] = 0.f;
    }
  }
}

}  // namespace
}  // namespace wn
