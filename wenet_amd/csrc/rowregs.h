// One wave per row with the row in registers, and the LayerNorm over it: shared by the
// row kernels of encoder_kernels.hip and the prologues that do their work inside a GEMM
// (gemm_x6r.hip DWC).  A private copy per translation unit (-fno-gpu-rdc).
#pragma once
#include "common.h"

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

}  // namespace
}  // namespace wn
