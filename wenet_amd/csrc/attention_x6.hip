// Six-product (bf16 x 3 planes) rel-pos self attention of the fp32 mode.  Built in round 4 (visits
// r07m-r07t, profiles/r07s_attention_x6_experiment.txt), kept out of the library then because it
// moves one utterance of the 256-utterance 8-rank golden across a pruning tie; re-landed in round
// 5 for the full-context encoders with that utterance proven a legitimate tie by the oracle
// (tests/test_gpu_dist.py) -- tune attn_x6.
// Rel-pos self attention (d_k = 64) of the fp32 mode with BOTH contractions as six bf16 plane
// products (gemm_x6.hip's arithmetic: every fp32 operand split exactly into three bf16 planes,
// the six products above 2^-26 accumulated in fp32) -- the twin of attention_kernel<NW, false,
// 2, true> (encoder_kernels.hip), which spends 4096 matrix-pipe cycles per 32-key tile and wave
// on v_mfma_f32_32x32x2_f32; here the same tile is 48 x v_mfma_f32_32x32x16_bf16 = 1536 cycles.
// Replaces RelPositionMultiHeadedAttention.forward (attention.py:364-438) for the encoder's
// full-context / chunk-masked self attention; rel-pos term folded into the keys as there:
//     (q + u) . k_j + (q + v) . p_j  =  q . (k_j + p_j)  +  (u . k_j + v . p_j).
//
// Two launches per layer:
//  1. attn_x6_pack_kernel, one block per (sequence, head, 32-key tile): K rows + position rows
//     (k' = k + p in fp32, per-key scalar u.k + v.p) and V rows -> split3 -> the tile's IMAGE in
//     HBM: three planes of K' [32 keys][64 dims], three of V^T [64 dims][32 key slots] (the
//     layout the fragment reads want) and the 32 scalars, 24.25 KB.  Every q block of a
//     (sequence, head) reads the same key tiles: folding and splitting them inside the
//     attention kernel (the first form of this file, r07m) repeated that work four to five
//     times and made the kernel VALU-issue bound (750 instructions per tile and wave).
//  2. attention_x6_kernel: one block = NW query groups x 2 key halves (waves [0, NW): first half
//     of the block's key tiles, waves [NW, 2 NW): second half, merged through LDS at the end),
//     one wave = 32 queries.  Per 32-key tile:
//   staging  the two halves' images, 16-byte chunks global -> registers -> LDS (no arithmetic)
//   S^T = K' Q^T      4 k blocks x 6 products; lane l holds, for ITS query (l & 31), the keys
//                     (r&3) + 8(r>>2) + 4(l>>5), r = 0..15
//   online softmax    on the raw scores (+ the per-key scalar), exp2 with scale * log2(e) folded
//                     into one fma per score
//   O^T += V^T P^T    the lane's probabilities, split into three planes in registers, ARE the B
//                     fragments (k slot (hi, e) <-> key 16j + 4hi + (e&3) + 8(e>>2), the slot
//                     order of the V^T image); 2 k blocks x 2 dim halves x 6 products.  O stays
//                     transposed (lane = query): rescale and 1 / l are the lane's own scalars.
#include "kernels.h"
#include "attn_x6_img.h"
#include "gemm_epilogue.h"
#include "tune.h"
#include "x6.h"

namespace wn {

namespace {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int KT = AX_KT;            // keys per tile
constexpr int KSTR = 72;             // K' plane row stride (bf16) in LDS: 64 dims + 16 B pad
constexpr int VSTR = AX_VSTR;        // V^T plane row stride (bf16): 32 key slots = four 16-B groups,
                                     // no pad: group g of dim row d lies at g ^ ((d >> 2) & 3)
constexpr int KPL = KT * KSTR;       // one K' plane (bf16 elements)
constexpr int VPL = AX_VPL;          // one V^T plane
constexpr int HALF = 3 * KPL + 3 * VPL;   // one key half's tile: 26112 B; a block: 52.5 KB,
                                          // three blocks per CU -- the 531 blocks of config 2
                                          // then run in ONE round (with two per CU = 512 slots the
                                          // last 19 started when the first ones ended: 43 us)
// the tile image in HBM (attn_x6_img.h): K' planes as [32][64] rows of 128 B (no pad), V^T planes
// exactly as in LDS, then the per-key scalars
constexpr int IMG_K = AX_IMG_K, IMG_V = AX_IMG_V, IMG_BIAS = AX_IMG_BIAS, IMG_TILE = AX_IMG_TILE;
// first tile of sequence s in the image: every sequence adds at most one partial tile
__device__ __forceinline__ int64_t img_tile0(int kvoff, int s) { return (kvoff >> 5) + s; }

__device__ __forceinline__ int vt_off(int d, int slot) { return ax_vt_off(d, slot); }

__device__ __forceinline__ float vmax(float a, float b) {   // bare v_max_f32 (attention_bf16.hip)
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// ---- launch 1: the tile images -----------------------------------------------------------------
// grid (tiles of the longest sequence, heads, sequences), 256 threads.  K' (and P): 32 rows x 16
// float4 chunks, natural mapping (coalesced rows); V: 16 key pairs x 16 float4 chunks: an item
// loads the same 4 dims of keys 2m, 2m + 1 (adjacent slots of V^T) and writes 4 packed bf16 pairs
// per plane.  The image is assembled in LDS and leaves as 16-byte chunks.
// GAL: tiles = the global 32-row blocks of the K / V matrix (AttnArgs::x6_galign): grid (row
// blocks, heads); a row's position row is its index inside ITS sequence (row_utt)
template <bool GAL>
__global__ __launch_bounds__(256) void attn_x6_pack_kernel(AttnArgs a, char* img) {
  const int s = GAL ? 0 : blockIdx.z, h = blockIdx.y, t = blockIdx.x;
  const int kvlen = GAL ? 0 : a.kv_len[s];
  if (!GAL && t * KT >= kvlen) return;
  const int kvoff = GAL ? 0 : a.kv_off[s];
  const int p_off = (!GAL && a.p_off) ? a.p_off[s] : 0;
  const int tid = threadIdx.x;
  __shared__ __attribute__((aligned(16))) char tile[IMG_TILE];
  const f32x4 fu = *reinterpret_cast<const f32x4*>(a.bias_u + h * 64 + (tid & 15) * 4);
  const f32x4 fv = *reinterpret_cast<const f32x4*>(a.bias_v + h * 64 + (tid & 15) * 4);
  f32x4 rK[2], rP[2], rV0, rV1;
  // row of the K / V matrix and row of the position table of tile-local key r
  auto krow = [&](int r, int* prow) {
    if constexpr (GAL) {
      const int g = min(t * KT + r, a.x6_rows - 1);
      const int u = a.row_utt[g];
      *prow = g - a.kv_off[u] + (a.p_off ? a.p_off[u] : 0);
      return g;
    } else {
      const int j = min(t * KT + r, kvlen - 1);
      *prow = j + p_off;
      return kvoff + j;
    }
  };
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + i * 256;
    const int r = c >> 4, c4 = c & 15;
    int pr;
    const int g = krow(r, &pr);
    rK[i] = *reinterpret_cast<const f32x4*>(a.K + (int64_t)g * a.ldk + h * 64 + c4 * 4);
    rP[i] = *reinterpret_cast<const f32x4*>(a.P + (int64_t)pr * a.ldp + h * 64 + c4 * 4);
  }
  {
    const int m = tid & 15, c4 = tid >> 4;
    int pr;
    const int g0 = krow(2 * m, &pr), g1 = krow(2 * m + 1, &pr);
    rV0 = *reinterpret_cast<const f32x4*>(a.V + (int64_t)g0 * a.ldv + h * 64 + c4 * 4);
    rV1 = *reinterpret_cast<const f32x4*>(a.V + (int64_t)g1 * a.ldv + h * 64 + c4 * 4);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + i * 256;
    const int r = c >> 4, c4 = c & 15;
    const f32x4 p = rP[i];
    f32x4 k = rK[i];
    float d = x6_key_scalar4(fu, k, fv, p);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    if (c4 == 0) reinterpret_cast<float*>(tile + IMG_BIAS)[r] = d;
    k = k + p;
    bf16x4 k0, k1, k2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const Split3 sp = split3(k[e]);
      k0[e] = sp.h0; k1[e] = sp.h1; k2[e] = sp.h2;
    }
    char* base = tile + r * 128 + c4 * 8;
    *reinterpret_cast<bf16x4*>(base) = k0;
    *reinterpret_cast<bf16x4*>(base + IMG_K) = k1;
    *reinterpret_cast<bf16x4*>(base + 2 * IMG_K) = k2;
  }
  {
    const int m = tid & 15, c4 = tid >> 4;
    const int slot = ax_key_slot(2 * m);               // tile-local key of rV0 (even)
    __bf16* vt = reinterpret_cast<__bf16*>(tile + 3 * IMG_K);   // key k + 1: slot + 1
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) {
      const Split3 t0 = split3(rV0[dd]), t1 = split3(rV1[dd]);
      bf16x2 p0, p1, p2;
      p0[0] = t0.h0; p0[1] = t1.h0;
      p1[0] = t0.h1; p1[1] = t1.h1;
      p2[0] = t0.h2; p2[1] = t1.h2;
      const int o = vt_off(c4 * 4 + dd, slot);
      *reinterpret_cast<bf16x2*>(vt + o) = p0;
      *reinterpret_cast<bf16x2*>(vt + VPL + o) = p1;
      *reinterpret_cast<bf16x2*>(vt + 2 * VPL + o) = p2;
    }
  }
  __syncthreads();
  char* dst = img + (((GAL ? 0 : img_tile0(kvoff, s)) + t) * a.n_heads + h) * IMG_TILE;
  for (int c = tid; c < IMG_TILE / 16; c += 256)
    *reinterpret_cast<f32x4*>(dst + c * 16) = *reinterpret_cast<const f32x4*>(tile + c * 16);
}

// ---- launch 2 ----------------------------------------------------------------------------------
// ABL (WN_ABLATION builds only, tune attn_x6_var; wrong results by design -- they attribute the
// kernel's time): 1 = tiles staged once, 2 = no MFMAs, 4 = no softmax arithmetic (exp2 / split3),
// 8 = no barriers in the loop (with 1)
template <int NW, int ABL = 0>
__global__ __launch_bounds__(NW * 2 * 64, 3) void attention_x6_kernel(AttnArgs a, const char* img) {
  int s = blockIdx.z, h = blockIdx.y, qb = blockIdx.x;
  if (a.blk_tab) {
    const int e = a.blk_tab[blockIdx.x];
    s = e >> 16; h = (e >> 8) & 255; qb = e & 255;
  } else if (a.xcd_nqb > 0) {
    // all query blocks of one (sequence, head) on the same XCD: its 4-5 blocks re-read the same
    // key-tile images (24 KB per tile), which then come from HBM / the fabric into ONE L2
    // instead of five (round 6: the launch fetched 126 MB for a 26-MB image)
    const int bid = xcd_block_order(blockIdx.x, gridDim.x);
    qb = bid % a.xcd_nqb;
    h = (bid / a.xcd_nqb) % a.n_heads;
    s = bid / (a.xcd_nqb * a.n_heads);
  }
  const int q0 = qb * (NW * 32);
  const int qlen = a.q_len[s];
  if (q0 >= qlen) return;
  const int kvlen = a.kv_len[s];
  const int qoff = a.q_off[s], kvoff = a.kv_off[s];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wave_all % NW;   // query group
  const int kh = wave_all / NW;     // key half
  const int hi = lane >> 5, li = lane & 31;
  constexpr int NTHR = NW * 2 * 64;

  __shared__ __attribute__((aligned(16))) __bf16 stile[2 * HALF];
  __shared__ __attribute__((aligned(16))) float sbias[2][KT];

  // ---- this lane's query row, dims kk*16 + hi*8 .. +7, as three planes --------------------
  const int qi = q0 + wave * 32 + li;
  const int qc = qi < qlen ? qi : qlen - 1;
  bf16x8 qp[3][4];
  {
    const float* qrow = a.Q + (int64_t)(qoff + qc) * a.ldq + h * 64 + hi * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(qrow + kk * 16);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(qrow + kk * 16 + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const Split3 t = split3(e < 4 ? x0[e] : x1[e - 4]);
        qp[0][kk][e] = t.h0; qp[1][kk][e] = t.h1; qp[2][kk][e] = t.h2;
      }
    }
  }
  // key window of this query: [jmin, jmax)
  int jmin = 0, jmax = kvlen;
  if (a.mask_mode == 1) {
    jmax = min(kvlen, qi + 1);
  } else if (a.mask_mode == 2) {
    const int c = qi / a.chunk_size;
    jmax = min(kvlen, (c + 1) * a.chunk_size);
    if (a.left_chunks >= 0) jmin = max((c - a.left_chunks) * a.chunk_size, 0);
  }
  // key range of the whole block (uniform)
  int blo = 0, bhi = kvlen;
  {
    const int qlast = min(q0 + NW * 32, qlen) - 1;
    if (a.mask_mode == 1) {
      bhi = min(kvlen, qlast + 1);
    } else if (a.mask_mode == 2) {
      bhi = min(kvlen, (qlast / a.chunk_size + 1) * a.chunk_size);
      if (a.left_chunks >= 0)
        blo = max((q0 / a.chunk_size - a.left_chunks) * a.chunk_size, 0);
    }
  }
  // x6_galign: tile t = global rows 32 t .. of the K / V matrix; key j of this sequence sits in
  // tile (kvoff + j) / 32, slots outside [0, kvlen) belong to the neighbours and are masked
  const int gsh = a.x6_galign ? kvoff : 0;
  const int t_lo = (blo + gsh) / KT, t_hi = (bhi + gsh + KT - 1) / KT;
  const int n_it = (t_hi - t_lo + 1) / 2;   // half kh works on tile t_lo + kh * n_it + it
  const int t_last = (kvlen - 1 + gsh) / KT;   // (a tile past the sequence: its last one again)

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;   // running maximum of the RAW scores, running sum
  // a query group past the sequence's end (the last block of a sequence with an odd number of
  // 32-query tiles) only stages and meets the barriers: its SIMD time goes to the other
  // blocks of the CU (round 6: such a block cost as much as a full one)
  const bool wave_live = q0 + wave * 32 < qlen;

  // ---- staging: the two halves' images as 16-byte chunks; thread tid moves chunk tid of every
  // plane of both halves (K' rows get their 16-byte pad on the way, the V^T planes and the
  // scalars are copied as they lie): addresses are two tile pointers + compile-time offsets
  const char* img0 =
      img + (((a.x6_galign ? 0 : img_tile0(kvoff, s)) * a.n_heads) + h) * (int64_t)IMG_TILE;
  const int64_t tile_stride = (int64_t)a.n_heads * IMG_TILE;
  char* lds0 = reinterpret_cast<char*>(stile);
  const int k_dst = (tid >> 3) * (KSTR * 2) + (tid & 7) * 16;
  static_assert(NTHR == 256, "one 16-byte chunk of a 4-KB plane per thread");
  auto stage = [&](int it) {
    const char* src[2] = {img0 + min(t_lo + it, t_last) * tile_stride + tid * 16,
                          img0 + min(t_lo + n_it + it, t_last) * tile_stride + tid * 16};
    f32x4 ck[2][3], cv[2][3], cb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        ck[hf][pl] = *reinterpret_cast<const f32x4*>(src[hf] + pl * IMG_K);
        cv[hf][pl] = *reinterpret_cast<const f32x4*>(src[hf] + 3 * IMG_K + pl * IMG_V);
      }
    if (tid < 16) cb = *reinterpret_cast<const f32x4*>(src[tid >> 3] - tid * 16 + IMG_BIAS + (tid & 7) * 16);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        *reinterpret_cast<f32x4*>(lds0 + hf * HALF * 2 + pl * KPL * 2 + k_dst) = ck[hf][pl];
        *reinterpret_cast<f32x4*>(lds0 + hf * HALF * 2 + 3 * KPL * 2 + pl * IMG_V + tid * 16) = cv[hf][pl];
      }
    if (tid < 16) *reinterpret_cast<f32x4*>(&sbias[tid >> 3][(tid & 7) * 4]) = cb;
  };

  // product order of gemm_x6r.hip: smallest first
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
  const float cs = a.scale * 1.4426950408889634f;   // exp2((raw - m) * scale * log2 e)
  const __bf16* sK = stile + kh * HALF;
  const __bf16* sV = sK + 3 * KPL;
  for (int it = 0; it < n_it; ++it) {
    const int kt = t_lo + kh * n_it + it;   // this wave's tile (may be >= t_hi)
    const int j0 = kt * KT - gsh;           // sequence-local key of the tile's slot 0 (< 0: a neighbour's)
    // no register prefetch: three waves per SIMD (<= 168 VGPRs) cover the load latency
    if (!(ABL & 1) || it == 0) stage(it);
    if (!(ABL & 8) || it == 0) __syncthreads();   // both halves' tiles visible
    if (kt < t_hi && wave_live) {
      // ---- S^T tile -------------------------------------------------------------------
      f32x16 sc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = 0.f;
      {
        const __bf16* kf = sK + li * KSTR + hi * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          bf16x8 fk[3];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            fk[pl] = *reinterpret_cast<const bf16x8*>(kf + pl * KPL + kk * 16);
          if constexpr (ABL & 2) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) sc[pl] += (float)fk[pl][0] * (float)qp[pl][kk][0];
          } else {
#pragma unroll
            for (int q = 0; q < 6; ++q)
              sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk[PA[q]], qp[PB[q]][kk], sc, 0, 0, 0);
          }
        }
      }
      // the per-key scalar: keys (r&3) + 8(r>>2) + 4hi -> four 16-byte reads
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(&sbias[kh][8 * g + 4 * hi]);
#pragma unroll
        for (int e = 0; e < 4; ++e) sc[4 * g + e] += b[e];
      }
      // ---- online softmax on this lane's query -------------------------------------
      const bool full = a.mask_mode == 0 && j0 >= 0 && j0 + KT <= kvlen;
      float psum = 0.f, alpha;
      if constexpr (ABL & 4) {
        alpha = 1.0f;
        psum = sc[0];
      } else if (full) {
        const float t0 = vmax3(sc[0], sc[1], sc[2]), t1 = vmax3(sc[3], sc[4], sc[5]);
        const float t2 = vmax3(sc[6], sc[7], sc[8]), t3 = vmax3(sc[9], sc[10], sc[11]);
        const float t4 = vmax3(sc[12], sc[13], sc[14]);
        float tmax = vmax(vmax3(t0, t1, t2), vmax3(t3, t4, sc[15]));
        tmax = vmax(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = vmax(m_run, tmax);
        alpha = __builtin_amdgcn_exp2f((m_run - m_new) * cs);
        const float mc = -m_new * cs;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], cs, mc));
        psum = ((sc[0] + sc[1]) + (sc[2] + sc[3])) + ((sc[4] + sc[5]) + (sc[6] + sc[7])) +
               (((sc[8] + sc[9]) + (sc[10] + sc[11])) + ((sc[12] + sc[13]) + (sc[14] + sc[15])));
        m_run = m_new;
      } else {
        float tmax = -1e30f;
        bool ok[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          ok[r] = (j >= jmin) && (j < jmax);
          if (ok[r]) tmax = vmax(tmax, sc[r]);
        }
        tmax = vmax(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = vmax(m_run, tmax);
        alpha = __builtin_amdgcn_exp2f((m_run - m_new) * cs);
        const float mc = -m_new * cs;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = ok[r] ? __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], cs, mc)) : 0.f;
          sc[r] = p;
          psum += p;
        }
        m_run = m_new;
      }
      l_run = l_run * alpha + psum;
      if (!__all(alpha == 1.0f)) {   // O^T: the factor is the lane's own
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          o0[r] *= alpha;
          o1[r] *= alpha;
        }
      }
      // ---- O^T += V^T P^T ----------------------------------------------------------------
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bf16x8 pa[3];
        if constexpr (ABL & 4) {
#pragma unroll
          for (int e = 0; e < 8; ++e) pa[0][e] = pa[1][e] = pa[2][e] = (__bf16)sc[8 * j + e];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const Split3 t = split3(sc[8 * j + e]);
            pa[0][e] = t.h0; pa[1][e] = t.h1; pa[2][e] = t.h2;
          }
        }
        bf16x8 v0[3], v1[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          v0[pl] = *reinterpret_cast<const bf16x8*>(sV + pl * VPL + vt_off(li, j * 16 + hi * 8));
          v1[pl] = *reinterpret_cast<const bf16x8*>(sV + pl * VPL + vt_off(32 + li, j * 16 + hi * 8));
        }
        if constexpr (ABL & 2) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            o0[pl] += (float)v0[pl][0] * (float)pa[pl][0];
            o1[pl] += (float)v1[pl][0] * (float)pa[pl][0];
          }
        } else {
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0[PA[q]], pa[PB[q]], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1[PA[q]], pa[PB[q]], o1, 0, 0, 0);
          }
        }
      }
    }
    if (!(ABL & 8)) __syncthreads();   // every wave is done with the tiles
  }
  float l_tot = l_run + __shfl_xor(l_run, 32, 64);

  // ---- merge the two key halves: half 1 parks (m, l, O^T) in LDS, half 0 folds it in.  Lane
  // (query li, hi) holds dims (r&3) + 8(r>>2) + 4hi (o0) / 32 + ... (o1) in BOTH halves.
  {
    float* xm = reinterpret_cast<float*>(stile) + wave * (32 * 65 + 64);   // m[32] l[32] O[32][65]
    float* xo = xm + 64;
    if (kh == 1) {
      if (hi == 0) { xm[li] = m_run; xm[32 + li] = l_tot; }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = (r & 3) + 8 * (r >> 2) + 4 * hi;
        xo[li * 65 + d] = o0[r];
        xo[li * 65 + 32 + d] = o1[r];
      }
    }
    __syncthreads();
    if (kh == 1) return;
    const float m1 = xm[li], l1 = xm[32 + li];
    const float m = vmax(m_run, m1);
    const float a0 = __builtin_amdgcn_exp2f((m_run - m) * cs);
    const float a1 = __builtin_amdgcn_exp2f((m1 - m) * cs);
    l_tot = l_tot * a0 + l1 * a1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = (r & 3) + 8 * (r >> 2) + 4 * hi;
      o0[r] = o0[r] * a0 + xo[li * 65 + d] * a1;
      o1[r] = o1[r] * a0 + xo[li * 65 + 32 + d] * a1;
    }
  }
  // ---- normalise and store: four consecutive dims per register quad ------------------------
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;   // fully-masked row -> 0
  if (qi < qlen) {
    float* op = a.O + (int64_t)(qoff + qi) * a.ldo + h * 64;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = 8 * g + 4 * hi;
      *reinterpret_cast<f32x4*>(op + d) =
          f32x4{o0[4 * g], o0[4 * g + 1], o0[4 * g + 2], o0[4 * g + 3]} * inv;
      *reinterpret_cast<f32x4*>(op + 32 + d) =
          f32x4{o1[4 * g], o1[4 * g + 1], o1[4 * g + 2], o1[4 * g + 3]} * inv;
    }
  }
}

}  // namespace

size_t attention_x6_image_bytes(int rows, int n_seq, int n_heads) {
  return ((size_t)(rows >> 5) + n_seq + 1) * n_heads * IMG_TILE;
}

bool attention_x6_supported(const AttnArgs& a) {
  // the folded rel-pos form over sequences long enough for the key split (the encoder's self
  // attention), with a scratch image from the caller; everything else stays on attention_kernel
  return a.P != nullptr && a.fold && a.bias_u && a.bias_v && a.kbias == nullptr &&
         a.max_q_len >= 128 && !a.o_bf16 && !a.qkv_bf16 && a.ldo % 4 == 0 && a.x6_img != nullptr &&
         a.q_off == a.kv_off && a.q_len == a.kv_len && (!a.x6_galign || a.row_utt != nullptr) &&
         (!a.x6_img_ready || a.x6_galign) &&
         a.x6_img_bytes >= attention_x6_image_bytes(a.x6_rows, a.n_seq, a.n_heads);
}

int attention_x6(const AttnArgs& a, hipStream_t s) {
  constexpr int NW = 2;
  char* img = reinterpret_cast<char*>(a.x6_img);
  if (a.x6_galign) {
    if (!a.x6_img_ready)
      hipLaunchKernelGGL(attn_x6_pack_kernel<true>, dim3(cdiv(a.x6_rows, KT), a.n_heads),
                         dim3(256), 0, s, a, img);
  } else {
    dim3 gp(cdiv(a.max_q_len, KT), a.n_heads, a.n_seq);
    hipLaunchKernelGGL(attn_x6_pack_kernel<false>, gp, dim3(256), 0, s, a, img);
  }
  dim3 g(cdiv(a.max_q_len, NW * 32), a.n_heads, a.n_seq), t(NW * 2 * 64);
  AttnArgs b = a;
  if (a.blk_tab && a.n_blk > 0) {
    g = dim3(a.n_blk);
  } else if (tune().attn_xcd != 0) {
    b.xcd_nqb = g.x;
    g = dim3(g.x * g.y * g.z);
  }
#ifdef WN_ABLATION
  switch (tune().attn_x6_var) {
    case 1: hipLaunchKernelGGL((attention_x6_kernel<NW, 1>), g, t, 0, s, b, img); break;
    case 2: hipLaunchKernelGGL((attention_x6_kernel<NW, 2>), g, t, 0, s, b, img); break;
    case 4: hipLaunchKernelGGL((attention_x6_kernel<NW, 4>), g, t, 0, s, b, img); break;
    case 6: hipLaunchKernelGGL((attention_x6_kernel<NW, 6>), g, t, 0, s, b, img); break;
    case 7: hipLaunchKernelGGL((attention_x6_kernel<NW, 7>), g, t, 0, s, b, img); break;
    case 9: hipLaunchKernelGGL((attention_x6_kernel<NW, 9>), g, t, 0, s, b, img); break;
    case 15: hipLaunchKernelGGL((attention_x6_kernel<NW, 15>), g, t, 0, s, b, img); break;
    default: hipLaunchKernelGGL((attention_x6_kernel<NW>), g, t, 0, s, b, img);
  }
#else
  hipLaunchKernelGGL((attention_x6_kernel<NW>), g, t, 0, s, b, img);
#endif
  WN_HIP(hipGetLastError());
  return 0;
}

}  // namespace wn
