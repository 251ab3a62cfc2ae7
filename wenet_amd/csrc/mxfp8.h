// OCP MXFP8 (e4m3 elements, one shared E8M0 scale per 32 consecutive k) helpers
// shared by the producers (LayerNorm, the GEMM epilogue, the weight quantiser) and
// restated bit for bit by the CPU checker used in the tests.
//
// Block scale rule (documented here because OCP leaves it to the implementation):
// the smallest power of two 2^e with amax <= 448 * 2^e (448 = 1.75 * 2^8 is the
// largest e4m3 value), so no element saturates:
//   amax = m * 2^x, 1 <= m < 2   ->   e = x - 8 + (m > 1.75)
// stored biased (E = e + 127), clamped to [0, 253]; elements = RNE_e4m3(v * 2^-e)
// (the multiplication by a power of two is exact).
#pragma once
#include <hip/hip_runtime.h>

namespace wn {

#ifdef __HIPCC__
__device__ __forceinline__ int mx_e8m0(float amax) {
  const unsigned bits = __float_as_uint(amax);
  const int x = (int)((bits >> 23) & 0xff);
  const int E = x - 8 + ((bits & 0x7fffffu) > 0x600000u ? 1 : 0);
  return min(max(E, 0), 253);
}
__device__ __forceinline__ float mx_inv_scale(int E) {   // 2^-(E - 127)
  return __uint_as_float((unsigned)(254 - E) << 23);
}
__device__ __forceinline__ int mx_pack4(float a, float b, float c, float d) {
  int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  return __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
}
#endif

}  // namespace wn
