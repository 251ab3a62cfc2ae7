// Shared pieces of the plane ("X3") operand format of gemm_x6.hip: the exact three-way
// bf16 split of an fp32 value and the record geometry (see gemm_x6.hip for the layout).
#pragma once
#include "common.h"

namespace wn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int X3_REC = 1024;            // one (k block, 32-row tile, plane) record
constexpr int X3_TILE = 3 * X3_REC;     // the three planes of a tile and k block

// x = h0 + h1 + h2 exactly: h0 = bf16(x) (round to nearest even), h1 = bf16(x - h0),
// h2 = x - h0 - h1 (<= 8 significant bits left, so the last conversion is exact; only
// below |x| ~ 2^-108, where h2 would be a subnormal bf16, is the sum off -- by < 2^-133)
struct Split3 { __bf16 h0, h1, h2; };
__device__ __forceinline__ Split3 split3(float x) {
  Split3 s;
  s.h0 = (__bf16)x;
  const float r1 = x - (float)s.h0;
  s.h1 = (__bf16)r1;
  s.h2 = (__bf16)(r1 - (float)s.h1);
  return s;
}

// One 4-dim piece of the per-key scalar u . k + v . p of the folded rel-pos attention
// (attention_x6.hip): ONE explicit order, shared by the pack pass and the QKV epilogue that
// replaces it, so that both give the same bits
__device__ __forceinline__ float x6_key_scalar4(const f32x4& fu, const f32x4& k, const f32x4& fv,
                                                const f32x4& p) {
  float d = fu[0] * k[0];
  d = __builtin_fmaf(fu[1], k[1], d);
  d = __builtin_fmaf(fu[2], k[2], d);
  d = __builtin_fmaf(fu[3], k[3], d);
  d = __builtin_fmaf(fv[0], p[0], d);
  d = __builtin_fmaf(fv[1], p[1], d);
  d = __builtin_fmaf(fv[2], p[2], d);
  d = __builtin_fmaf(fv[3], p[3], d);
  return d;
}

// byte offset of the 16-B piece (k half h) of image row `row`, k block kb, plane 0
__device__ __forceinline__ int64_t x3_piece(int kb, int tiles, int row, int h) {
  return ((int64_t)kb * tiles + (row >> 5)) * X3_TILE + h * 512 + (row & 31) * 16;
}

#ifdef __HIPCC__
// One 32-column tile of a matrix held the way the operand-swapped MFMA leaves it -- lane = row
// (li = lane & 31 of row tile `tile`), v[g] = columns 8 g + 4 hi .. + 3 of the tile, hi =
// lane >> 5 -- stored as its pieces of the X3 image `img` (`tiles` row tiles): exact split,
// then one half-wave exchange (v_permlane32_swap) so that the low lane holds columns 0-15 (k
// block kb0), the high lane 16-31 (k block kb0 + 1) of its row as whole 16-byte pieces: 512
// contiguous bytes per half-wave and store.  (gemm_x6.hip's EPI 2, shared by the row-block
// kernels that hand LN(x) to a six-product consumer as fragments, round 5.)
__device__ __forceinline__ void x3_store_tile(const f32x4 (&v)[4], void* img, int kb0, int tiles,
                                              int tile, int hi, int li) {
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  i32x2 q[3][4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    bf16x4 h0, h1, h2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const Split3 sp = split3(v[g][e]);
      h0[e] = sp.h0; h1[e] = sp.h1; h2[e] = sp.h2;
    }
    q[0][g] = __builtin_bit_cast(i32x2, h0);
    q[1][g] = __builtin_bit_cast(i32x2, h1);
    q[2][g] = __builtin_bit_cast(i32x2, h2);
  }
#pragma unroll
  for (int pl = 0; pl < 3; ++pl)
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      auto s0 = __builtin_amdgcn_permlane32_swap(q[pl][0][d], q[pl][2][d], false, false);
      auto s1 = __builtin_amdgcn_permlane32_swap(q[pl][1][d], q[pl][3][d], false, false);
      q[pl][0][d] = s0[0]; q[pl][2][d] = s0[1];
      q[pl][1][d] = s1[0]; q[pl][3][d] = s1[1];
    }
  char* o = reinterpret_cast<char*>(img) + ((int64_t)(kb0 + hi) * tiles + tile) * X3_TILE + li * 16;
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    // half 0: columns 0-7 of the k block = (own g0 | partner's g0), half 1: g1
    *reinterpret_cast<i32x4*>(o + pl * X3_REC) =
        i32x4{q[pl][0][0], q[pl][0][1], q[pl][2][0], q[pl][2][1]};
    *reinterpret_cast<i32x4*>(o + pl * X3_REC + 512) =
        i32x4{q[pl][1][0], q[pl][1][1], q[pl][3][0], q[pl][3][1]};
  }
}
#endif

}  // namespace wn
