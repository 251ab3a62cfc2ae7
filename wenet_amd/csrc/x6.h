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

// byte offset of the 16-B piece (k half h) of image row `row`, k block kb, plane 0
__device__ __forceinline__ int64_t x3_piece(int kb, int tiles, int row, int h) {
  return ((int64_t)kb * tiles + (row >> 5)) * X3_TILE + h * 512 + (row & 31) * 16;
}

}  // namespace wn
