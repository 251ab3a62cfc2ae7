// Geometry of the key-tile image of the six-product rel-pos attention (attention_x6.hip): shared
// by its pack pass, its kernel and the QKV projection's epilogue that writes the tiles directly
// (gemm_x6r.hip, EPI 4).  One tile = 32 keys of one head:
//   3 planes of K' = k + p     [32 keys][64 dims] bf16, rows of 128 B
//   3 planes of V^T            [64 dims][32 key slots] bf16, rows of 64 B, slot order and 16-B
//                              group swizzle as the attention kernel's LDS reads want them
//   32 per-key scalars u.k + v.p (fp32), padded to 256 B
#pragma once

namespace wn {

constexpr int AX_KT = 32;                               // keys per tile
constexpr int AX_VSTR = 32;                             // V^T plane row stride (bf16)
constexpr int AX_VPL = 64 * AX_VSTR;                    // one V^T plane (bf16 elements)
constexpr int AX_IMG_K = AX_KT * 64 * 2;                // 4096 B per K' plane
constexpr int AX_IMG_V = AX_VPL * 2;                    // 4096 B per V^T plane
constexpr int AX_IMG_BIAS = 3 * AX_IMG_K + 3 * AX_IMG_V;   // byte offset of the 32 scalars
constexpr int AX_IMG_TILE = AX_IMG_BIAS + 256;          // 24832 B

// key slot of tile-local key k: the order in which a lane of the attention kernel holds its
// probabilities (k slot (hi, e) <-> key 16 j + 4 hi + (e & 3) + 8 (e >> 2))
__host__ __device__ __forceinline__ int ax_key_slot(int k) {
  const int k16 = k & 15;
  return (k >> 4) * 16 + ((k16 >> 2) & 1) * 8 + (k16 & 3) + 4 * (k16 >> 3);
}

// element offset of key slot `slot` (0..31) of dim row d inside a V^T plane: 16-byte group g of
// dim row d lies at g ^ ((d >> 2) & 3)
__host__ __device__ __forceinline__ int ax_vt_off(int d, int slot) {
  return d * AX_VSTR + ((((slot >> 3) ^ (d >> 2)) & 3) << 3) + (slot & 7);
}

}  // namespace wn
