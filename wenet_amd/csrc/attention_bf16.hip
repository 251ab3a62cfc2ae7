// Multi-head attention (d_k = 64) with bf16 operands on the bf16 matrix cores --
// the WN_PREC_BF16 twin of attention_kernel (encoder_kernels.hip): same
// arguments, masks, online softmax in fp32 and fp32 output; Q (+ pos_bias_u/v),
// K, the projected position rows and the probabilities / V are rounded to bf16
// (RNE) where they enter an MFMA.  Under the reference's autocast
// (wenet/bin/recognize.py:278-280) torch.matmul(q, k^T) and torch.matmul(attn, v)
// (wenet/models/transformer/attention.py:133-178,364-438) run with bf16 operands
// as well; the softmax there and here is fp32.
//
// One block = NW waves, each wave owns 32 query rows of one (sequence, head);
// the block shares every 32-key K / V (/ P) tile, staged global (fp32) ->
// registers -> bf16 LDS one tile ahead (double-buffered, one barrier per tile).
// Per tile and wave:
//   S^T = K (Q+u)^T [+ P (Q+v)^T]   4 (8) x v_mfma_f32_32x32x16_bf16
//       lane l holds, for ITS query (l & 31), keys (r&3) + 8(r>>2) + 4(l>>5)
//   online softmax on the lane's 16 scores (max / sum: one exchange with lane^32)
//   O^T += V^T P^T                   4 x v_mfma_f32_32x32x16_bf16
//       the lane's probabilities r = 8j .. 8j+7 ARE the "B" fragment of MFMA j
//       (k slot (hi, e) <-> key 16j + 4hi + (e&3) + 8(e>>2)); V is stored
//       TRANSPOSED in LDS, Vt[dim][slot], slot = 16j + 8hi + e, so the "A" fragment
//       (8 keys of one dim) is one 16-byte read; the output stays transposed (lane =
//       query, registers = dims): the online-softmax rescale and the final 1 / l are the
//       lane's own scalars (round 3; before, O had the queries along the registers and
//       every rescale cost 16 cross-lane reads).
// With 16x the MFMA rate the softmax VALU work (16 scores per lane and tile)
// bounds the kernel; NW grows with the sequence length so that the fp32 K / V
// stream from L2 is shared by more queries (64 B/clk/CU budget).
#include <type_traits>

#include "gemm_epilogue.h"
#include "kernels.h"

namespace wn {


namespace {

// Maxima of the online softmax as the bare v_max_f32 / v_max3_f32 (IEEE maxNum: a NaN operand
// yields the other one; the NaN still reaches the output through exp2 and the row sum).  Plain
// fmaxf on an MFMA result makes the compiler put a canonicalising v_max x, x in front of every
// use unless the whole file is built with -fno-honor-nans (rounds 2-3; dropped in round 4).
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
template <typename V16>
__device__ __forceinline__ float vmax16(const V16& v) {
  const float a = vmax3(v[0], v[1], v[2]), b = vmax3(v[3], v[4], v[5]), c = vmax3(v[6], v[7], v[8]);
  const float d = vmax3(v[9], v[10], v[11]), e = vmax3(v[12], v[13], v[14]);
  return vmax(vmax3(a, b, c), vmax3(d, e, v[15]));
}

// The asm maxima above are invisible to the compiler's hazard recogniser: a VALU read of an MFMA
// result needs the XDL write-back wait states (s_nop 11 after v_mfma_f32_32x32x16_bf16, what the
// compiler itself puts in front of a plain VALU read).  Every softmax takes its scores through
// this first: the asm "redefines" the accumulator tuple, so all later reads -- asm or not --
// are ordered behind the wait.
template <typename V16>
__device__ __forceinline__ void mfma_settle(V16& v) {
  asm("s_nop 11" : "+v"(v));
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int KT = 32;     // keys per tile
constexpr int KSTR = 72;   // K / P tile row stride (bf16): 64 dims + 16 B pad
constexpr int VSTR = 40;   // Vt row stride (bf16): 32 key slots + 16 B pad

__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  bf16x8 r;
  r[0] = (__bf16)a[0]; r[1] = (__bf16)a[1]; r[2] = (__bf16)a[2]; r[3] = (__bf16)a[3];
  r[4] = (__bf16)b[0]; r[5] = (__bf16)b[1]; r[6] = (__bf16)b[2]; r[7] = (__bf16)b[3];
  return r;
}

// SUB: 32-key sub-tiles staged per barrier (2 doubles the MFMA work and the prefetch lead
// per __syncthreads for long sequences); IN16: Q / K / V are bf16 matrices in HBM (written
// by the QKV GEMM of the bf16-storage form; ld* in bf16 elements) -- half the K / V
// stream, no conversion on the way into LDS.  Same arithmetic: the kernel rounds Q, K, V
// to bf16 first thing anyway.
template <int NW, bool RELPOS, int SUB = 1, bool IN16 = false>
__global__ __launch_bounds__(NW * 64, (RELPOS || NW == 2) ? 2 : 4) void attention_bf16_kernel(AttnArgs a) {
  static_assert(!(IN16 && RELPOS), "bf16 Q/K/V: plain attention only (q + bias_u is rounded once)");
  const int s = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * (NW * 32);
  const int qlen = a.q_len[s];
  if (q0 >= qlen) return;
  const int kvlen = a.kv_len[s];
  const int qoff = a.q_off[s], kvoff = a.kv_off[s];
  const int p_off = a.p_off ? a.p_off[s] : 0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, li = lane & 31;
  constexpr int NTHR = NW * 64;

  // LDS: [2 buffers][K tile | Vt tile | P tile]
  constexpr int KMAT = KT * KSTR;        // bf16 elements
  constexpr int VMAT = 64 * VSTR;
  constexpr int BUF = KMAT + VMAT + (RELPOS ? KMAT : 0);
  __shared__ __attribute__((aligned(16))) __bf16 stile[2 * SUB * BUF];
  typedef typename std::conditional<IN16, bf16x4, f32x4>::type ld4_t;
  const __bf16* Qh = reinterpret_cast<const __bf16*>(a.Q);
  const __bf16* Kh = reinterpret_cast<const __bf16*>(a.K);
  const __bf16* Vh = reinterpret_cast<const __bf16*>(a.V);

  // ---- this lane's query row: dims kk*16 + hi*8 .. +7, kk = 0..3 ------------
  const int qi = q0 + wave * 32 + li;
  const int qc = qi < qlen ? qi : qlen - 1;
  bf16x8 qu[4], qv[RELPOS ? 4 : 1];
  if constexpr (IN16) {
    const __bf16* qp = Qh + (int64_t)(qoff + qc) * a.ldq + h * 64 + hi * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qu[kk] = *reinterpret_cast<const bf16x8*>(qp + kk * 16);
  } else {
    const float* qp = a.Q + (int64_t)(qoff + qc) * a.ldq + h * 64 + hi * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(qp + kk * 16);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(qp + kk * 16 + 4);
      if (RELPOS) {
        const float* bu = a.bias_u + h * 64 + kk * 16 + hi * 8;
        const float* bv = a.bias_v + h * 64 + kk * 16 + hi * 8;
        qu[kk] = pack8(x0 + *reinterpret_cast<const f32x4*>(bu),
                       x1 + *reinterpret_cast<const f32x4*>(bu + 4));
        qv[kk] = pack8(x0 + *reinterpret_cast<const f32x4*>(bv),
                       x1 + *reinterpret_cast<const f32x4*>(bv + 4));
      } else {
        qu[kk] = pack8(x0, x1);
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
  const int t_lo = blo / KT, t_hi = (bhi + KT - 1) / KT;
  const int n_it = (t_hi - t_lo + SUB - 1) / SUB;     // stages of SUB sub-tiles

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;      // running maximum of the RAW scores, running sum

  // ---- tile staging ---------------------------------------------------------
  // K (and P): 32 rows x 16 float4 chunks, natural mapping (coalesced rows).
  // V: 16 key pairs x 16 float4 chunks; an item loads the same 4 dims of keys
  // 2m, 2m+1 (adjacent slots of Vt) and writes 4 packed bf16 pairs.
  constexpr int NCK = (KT * 16 + NTHR - 1) / NTHR;   // K chunks per thread
  constexpr int NCV = (KT * 8 + NTHR - 1) / NTHR;    // V items per thread
  ld4_t rK[SUB][NCK], rV0[SUB][NCV], rV1[SUB][NCV];
  f32x4 rP[RELPOS ? NCK : 1];
  auto gload = [&](int it) {
#pragma unroll
    for (int sb = 0; sb < SUB; ++sb) {
      const int jt = (t_lo + it * SUB + sb) * KT;
#pragma unroll
      for (int i = 0; i < NCK; ++i) {
        const int c = tid + i * NTHR;
        if (KT * 16 % NTHR == 0 || c < KT * 16) {
          const int r = c >> 4, c4 = c & 15;
          int j = jt + r;
          if (j > kvlen - 1) j = kvlen - 1;
          if constexpr (IN16)
            rK[sb][i] = *reinterpret_cast<const bf16x4*>(Kh + (int64_t)(kvoff + j) * a.ldk +
                                                         h * 64 + c4 * 4);
          else
            rK[sb][i] = *reinterpret_cast<const f32x4*>(a.K + (int64_t)(kvoff + j) * a.ldk +
                                                        h * 64 + c4 * 4);
          if (RELPOS)
            rP[i] = *reinterpret_cast<const f32x4*>(a.P + (int64_t)(j + p_off) * a.ldp +
                                                    h * 64 + c4 * 4);
        }
      }
#pragma unroll
      for (int i = 0; i < NCV; ++i) {
        const int c = tid + i * NTHR;
        if (KT * 8 % NTHR == 0 || c < KT * 8) {
          const int m = c & 15, c4 = c >> 4;
          int j0 = jt + 2 * m, j1 = j0 + 1;
          if (j0 > kvlen - 1) j0 = kvlen - 1;
          if (j1 > kvlen - 1) j1 = kvlen - 1;
          if constexpr (IN16) {
            rV0[sb][i] = *reinterpret_cast<const bf16x4*>(Vh + (int64_t)(kvoff + j0) * a.ldv +
                                                          h * 64 + c4 * 4);
            rV1[sb][i] = *reinterpret_cast<const bf16x4*>(Vh + (int64_t)(kvoff + j1) * a.ldv +
                                                          h * 64 + c4 * 4);
          } else {
            rV0[sb][i] = *reinterpret_cast<const f32x4*>(a.V + (int64_t)(kvoff + j0) * a.ldv +
                                                         h * 64 + c4 * 4);
            rV1[sb][i] = *reinterpret_cast<const f32x4*>(a.V + (int64_t)(kvoff + j1) * a.ldv +
                                                         h * 64 + c4 * 4);
          }
        }
      }
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int sb = 0; sb < SUB; ++sb) {
      __bf16* base = stile + (buf * SUB + sb) * BUF;
#pragma unroll
      for (int i = 0; i < NCK; ++i) {
        const int c = tid + i * NTHR;
        if (KT * 16 % NTHR == 0 || c < KT * 16) {
          const int r = c >> 4, c4 = c & 15;
          bf16x4 k4;
          k4[0] = (__bf16)rK[sb][i][0]; k4[1] = (__bf16)rK[sb][i][1];
          k4[2] = (__bf16)rK[sb][i][2]; k4[3] = (__bf16)rK[sb][i][3];
          *reinterpret_cast<bf16x4*>(base + r * KSTR + c4 * 4) = k4;
          if (RELPOS) {
            bf16x4 p4;
            p4[0] = (__bf16)rP[i][0]; p4[1] = (__bf16)rP[i][1];
            p4[2] = (__bf16)rP[i][2]; p4[3] = (__bf16)rP[i][3];
            *reinterpret_cast<bf16x4*>(base + KMAT + VMAT + r * KSTR + c4 * 4) = p4;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NCV; ++i) {
        const int c = tid + i * NTHR;
        if (KT * 8 % NTHR == 0 || c < KT * 8) {
          const int m = c & 15, c4 = c >> 4;
          const int k = 2 * m;                       // tile-local key of rV0
          const int k16 = k & 15;
          const int slot = (k >> 4) * 16 + ((k16 >> 2) & 1) * 8 + (k16 & 3) +
                           4 * (k16 >> 3);          // even; key k+1 is slot+1
          __bf16* vt = base + KMAT + (c4 * 4) * VSTR + slot;
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            bf16x2 pr;
            pr[0] = (__bf16)rV0[sb][i][d];
            pr[1] = (__bf16)rV1[sb][i][d];
            *reinterpret_cast<bf16x2*>(vt + d * VSTR) = pr;
          }
        }
      }
    }
  };
  if (n_it > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();

  // exponent in the log2 domain: exp2((s - m) * scale * log2(e)), v_exp_f32 directly
  const float cs = a.scale * 1.4426950408889634f;
  for (int it = 0; it < n_it; ++it) {
    const int cur = it & 1;
    if (it + 1 < n_it) gload(it + 1);
#pragma unroll
    for (int sb = 0; sb < SUB; ++sb) {
      if (SUB > 1 && t_lo + it * SUB + sb >= t_hi) break;   // uniform
      const int j0 = (t_lo + it * SUB + sb) * KT;
      const __bf16* sK = stile + (cur * SUB + sb) * BUF;
      const __bf16* sV = sK + KMAT;
      const __bf16* sP = sK + KMAT + VMAT;

      // ---- S^T tile -----------------------------------------------------------
      f32x16 sc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = 0.f;
      {
        const __bf16* kf = sK + li * KSTR + hi * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const bf16x8 fk = *reinterpret_cast<const bf16x8*>(kf + kk * 16);
          sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk, qu[kk], sc, 0, 0, 0);
        }
        if (RELPOS) {
          const __bf16* pf = sP + li * KSTR + hi * 8;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const bf16x8 fp = *reinterpret_cast<const bf16x8*>(pf + kk * 16);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fp, qv[kk], sc, 0, 0, 0);
          }
        }
      }
      // ---- online softmax on this lane's query ----------------------------------
      // The running maximum is kept on the RAW scores (scale > 0: same arg-max); the scale and
      // log2(e) enter through ONE fma per score, exp2(s cs - m cs), instead of a multiply, a
      // subtract and the window select: the loop is bound by this VALU work (16 scores per
      // lane and 32-key tile against 8 MFMAs), not by the matrix pipe.  Interior tiles of an
      // unmasked sequence take the branch without any per-key window test (uniform).
      mfma_settle(sc);
      const bool full = a.mask_mode == 0 && j0 + KT <= kvlen;
      float psum = 0.f, alpha;
      if (full) {
        float tmax = vmax16(sc);
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
      // rescale the running output: O is kept TRANSPOSED (rows = dims, column = this lane's
      // query), so the factor is the lane's own -- no exchange between lanes
      if (!__all(alpha == 1.0f)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          o0[r] *= alpha;
          o1[r] *= alpha;
        }
      }
      // ---- O += P V ---------------------------------------------------------------
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bf16x8 pa;
#pragma unroll
        for (int e = 0; e < 8; ++e) pa[e] = (__bf16)sc[8 * j + e];
        const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(sV + li * VSTR + j * 16 + hi * 8);
        const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(sV + (32 + li) * VSTR + j * 16 +
                                                           hi * 8);
        // O^T += V^T P^T: the Vt fragment is the "A" operand (rows = dims), the lane's own
        // probabilities the "B" operand (column = its query)
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pa, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pa, o1, 0, 0, 0);
      }
    }
    if (it + 1 < n_it) lstore(cur ^ 1);
    __syncthreads();  // next stage visible, this buffer free
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  // ---- normalise and store ---------------------------------------------------
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;  // fully-masked row -> 0
  // lane = query row, registers = dims (r&3) + 8(r>>2) + 4 hi (o0) / 32 + ... (o1): four
  // consecutive dims per register quad
  const int qrow = q0 + wave * 32 + li;
  if (qrow < qlen) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = 8 * g + 4 * hi;
      const f32x4 a0 = f32x4{o0[4 * g], o0[4 * g + 1], o0[4 * g + 2], o0[4 * g + 3]} * inv;
      const f32x4 a1 = f32x4{o1[4 * g], o1[4 * g + 1], o1[4 * g + 2], o1[4 * g + 3]} * inv;
      if (a.o_bf16) {  // bf16-storage mode: the context only feeds the out-proj GEMM
        __bf16* op = reinterpret_cast<__bf16*>(a.O) + (int64_t)(qoff + qrow) * a.ldo + h * 64;
        bf16x4 b0, b1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { b0[e] = (__bf16)a0[e]; b1[e] = (__bf16)a1[e]; }
        *reinterpret_cast<bf16x4*>(op + d) = b0;
        *reinterpret_cast<bf16x4*>(op + 32 + d) = b1;
      } else {
        float* op = a.O + (int64_t)(qoff + qrow) * a.ldo + h * 64;
        *reinterpret_cast<f32x4*>(op + d) = a0;
        *reinterpret_cast<f32x4*>(op + 32 + d) = a1;
      }
    }
  }
}

// ---- DMA-staged form (round 3): bf16 Q | K rows of the QKV GEMM + a packed V^T image --------
// The register-staged kernel above spends ~45 of its ~135 VALU instructions per 32-key tile
// on moving K / V (64-bit addresses, bf16 pair packing, five LDS stores) and hipcc waits for
// the just-issued prefetch loads inside the first MFMAs of every stage.  Here a stage of 64
// keys is FOUR 1-KB direct-to-LDS pieces per wave pair (buffer_load ... lds, no registers, no
// VALU): K rows as they lie in the QKV matrix, and V from a V^T image [seq][head][dim][Tp]
// (vt_pack_kernel, one pass over V per layer) whose rows are key-contiguous, the 16 keys of
// a group already in the key order of the S^T registers.  Both tiles are [64 rows][128 B]
// with the 16-byte slot swizzle slot ^ ((row >> 1) & 7) applied on the SOURCE side of the DMA
// (the LDS image of a piece is lane-linear) and again on the fragment reads, as in
// gemm_bf16p.hip: a ds_read_b128 of 32 consecutive rows touches 16 distinct slots per
// service group.  Two stages of LDS (32 KB), one raw s_barrier per stage behind the issuing
// waves' own vmcnt(0); the next stage's pieces fly during the whole compute of this one.
// Same arithmetic, in the same order, as the register-staged kernel (bit-identical output).
constexpr int DKT = 64;                 // keys per stage
constexpr int DTILE = 64 * 128;         // one [64][128 B] tile
constexpr int DSTAGE = 2 * DTILE;       // K tile | V^T tile

__device__ __forceinline__ float pair_max(float v) {   // max over lanes l, l ^ 32
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v),
                                                  false, false);
  return vmax(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float pair_sum(float v) {   // sum over lanes l, l ^ 32
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v),
                                                  false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// PIPE (attn_bf16_dma = 2): the stage loop unrolled over the two LDS buffers so that every
// fragment address is ONE of four per-lane registers + an immediate, the K fragments of a
// sub-tile read as a group before its MFMA chain (the second sub-tile's, and the V^T fragments,
// under the softmax of the first) instead of read - wait - multiply one at a time.
// VTR (attn_bf16_dma = 4, with PIPE): no V^T image at all -- the V rows are staged by DMA exactly
// like the K rows ([64 keys][128 B], same swizzle) and the PV "A" fragments (8 keys of one dim per
// lane) come out of ds_read_b64_tr_b16, which hands lane j of a 16-lane group COLUMN j of the
// [4 keys][16 dims] block the group's lanes address (probed on hardware:
// tools/probes/tr16_probe.hip, profiles/r05z_tr16_probe.txt): two transpose reads per fragment
// (keys 16 j + 4 hi + 0..3 and + 8..11 -- the register order of the S^T tile), four per-lane
// offsets + immediates.  Saves the vt_pack pass (25 us and 123 MB per layer at config 5).
//
// TRA (attn_bf16_dma = 5, measurement form of VTR, not the default): the transpose reads as
// inline asm.  Behind an LDS-DMA in flight the compiler orders every LDS read it cannot prove
// disjoint from the DMA's destination with s_waitcnt vmcnt(0) (SIInsertWaitcnts: reads whose
// memory operand carries no alias scope wait for ALL LDS-DMA) -- the plain K / V^T reads escape
// that, the ds_read_tr16_b64 builtin does not: the VTR kernel's ISA has a vmcnt(0) in front of
// the first V read of every stage, i.e. the prefetch of stage it + 1 issued behind the barrier
// has to land under ONE score tile + softmax instead of a whole stage (found by reading the
// ISA, end of round 3; measured since: bit-identical, 0.2-0.3 % on config 5 -- the other waves of
// the SIMD cover that wait).  The reads of stage `it` touch
// buffer BUF only, the DMA in flight writes BUF ^ 1: no wait is needed.  An asm read is
// invisible to the compiler's lgkmcnt bookkeeping, so the waits are explicit (LDS operations
// of a wave return in order: a count that ignores younger compiler-issued reads is only ever
// too strict).
template <int OFF>
__device__ __forceinline__ short __attribute__((ext_vector_type(4))) tr16_b64_asm(unsigned addr) {
  short __attribute__((ext_vector_type(4))) r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

// Round 4: TRA is the only form built (the V^T-image forms attn_bf16_dma = 1 / 2 and the builtin
// transpose reads = 4 were bit-identical and slower or equal: removed; the flags below keep the
// kernel text readable against the descriptions above).
template <int NW>
__global__ __launch_bounds__(NW * 64, 4) void attention_bf16_dma_kernel(AttnArgs a, int nqb) {
  constexpr bool PIPE = true, VTR = true, TRA = true;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  static_assert(NW == 4 || NW == 8, "4 or 8 query groups per block");
  static_assert(!VTR || PIPE, "transpose-read form: grouped-read kernel only");
  static_assert(!TRA || VTR, "asm transpose reads: a form of the transpose-read kernel");
  constexpr int NP = 8 / NW;              // 1-KB pieces of a tile per wave
  __shared__ __attribute__((aligned(1024))) char sbuf[2 * DSTAGE];
  // all query blocks of one (sequence, head) run on the same XCD, one after the other: its
  // K / V^T rows (2 x 192 KB at T = 1500) come from HBM into one L2 instead of eight
  const int bid = xcd_block_order(blockIdx.x, gridDim.x);
  const int qb = bid % nqb;
  const int h = (bid / nqb) % a.n_heads, s = bid / (nqb * a.n_heads);
  const int q0 = qb * (NW * 32);
  const int qlen = a.q_len[s];
  if (q0 >= qlen) return;
  const int kvlen = a.kv_len[s];
  const int qoff = a.q_off[s], kvoff = a.kv_off[s];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, li = lane & 31;
  const __bf16* Qh = reinterpret_cast<const __bf16*>(a.Q);

  // ---- DMA descriptors and per-lane source offsets ------------------------------------
  const unsigned ldk2 = (unsigned)a.ldk * 2u;
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(reinterpret_cast<const __bf16*>(a.K) + (int64_t)kvoff * a.ldk + h * 64),
      0, (int)((unsigned)kvlen * ldk2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(reinterpret_cast<const __bf16*>(a.V) + (int64_t)kvoff * a.ldv + h * 64),
      0, (int)((unsigned)kvlen * ldk2), 0x00020000);          // ldv == ldk (checked by the host)
  int prow[NP];
  unsigned pslot[NP], vtoff[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    prow[p] = (p * NW + wave) * 8 + (lane >> 3);                       // row of the tile
    pslot[p] = (unsigned)(((lane & 7) ^ ((prow[p] >> 1) & 7)) << 4);   // source 16-B slot
    // VTR: the V tile has its own swizzle -- slot bit 2 flipped on rows 2, 3 (mod 4): a half-wave
    // of a transpose read touches 4 consecutive rows x 64 B, and with the K swizzle rows r and
    // r + 2 of them met in the same banks (SQ_LDS_BANK_CONFLICT 11.6 M cycles per launch, r05ad);
    // this field is then the V source slot minus the K one
    vtoff[p] = (unsigned)((((lane & 7) ^ (((prow[p] >> 1) & 1) << 2)) << 4)) - pslot[p];
  }
  auto issue = [&](int it, int buf) {
    const int j0 = it * DKT;
    char* dst = sbuf + buf * DSTAGE + wave * 1024;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const unsigned vk = (unsigned)min(j0 + prow[p], kvlen - 1) * ldk2 + pslot[p];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_ptr)(dst + p * NW * 1024), 16, vk, 0, 0,
                                               0);
      const unsigned vv = vk + vtoff[p];         // the same rows of V, its own slot swizzle
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr)(dst + DTILE + p * NW * 1024), 16,
                                               vv, 0, 0, 0);
    }
  };
  const int n_it = (kvlen + DKT - 1) / DKT;
  issue(0, 0);

  // ---- this lane's query row: dims kk*16 + hi*8 .. +7, kk = 0..3 ----------------------
  const int qi = q0 + wave * 32 + li;
  const int qc = qi < qlen ? qi : qlen - 1;
  bf16x8 qu[4];
  {
    const __bf16* qp = Qh + (int64_t)(qoff + qc) * a.ldq + h * 64 + hi * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qu[kk] = *reinterpret_cast<const bf16x8*>(qp + kk * 16);
  }
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;
  const float cs = a.scale * 1.4426950408889634f;
  // fragment addresses: row li of a 32-row half (the swizzle term (row >> 1) & 7 is the same
  // for rows li and 32 + li), 16-B slot c -> c ^ sw
  const int sw = (li >> 1) & 7;
  const int frow = li * 128;

  if constexpr (PIPE) {
    // per-lane fragment offsets: 16-B slot 2 c + hi of row li, c = 0..3 (K: k step c; V^T: key
    // group c = 2 sb + j)
    int fo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) fo[c] = frow + (((2 * c + hi) ^ sw) << 4);
    // VTR: this lane's address inside the [4 keys][16 dims] block its 16-lane group reads, for
    // the first / second key quad of a fragment (w) and the low / high 32 dims (dblk): key row
    // 4 hi + 8 w + (j' >> 2) of the 16-key group, dims 32 dblk + 16 (g & 1) + 4 (j' & 3) .. + 3
    // (the 16-key groups start at multiples of 16 rows: the swizzle term depends on this row only)
    int tro[2][2];
    if constexpr (VTR) {
      const int jj = lane & 15, g1 = (lane >> 4) & 1;
#pragma unroll
      for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int dblk = 0; dblk < 2; ++dblk) {
          const int row = 4 * hi + 8 * w + (jj >> 2);
          const int swz = ((row >> 1) & 1) << 2;           // the V tile's swizzle (see issue)
          const int slot = 4 * dblk + 2 * g1 + ((jj & 3) >> 1);
          tro[w][dblk] = row * 128 + ((slot ^ swz) << 4) + (jj & 1) * 8;
        }
    }
    // TRA: the same four offsets as LDS byte addresses (buffer, tile and key group go into the
    // instruction's immediate)
    unsigned tra[2][2] = {{0u, 0u}, {0u, 0u}};
    if constexpr (TRA) {
      const unsigned base = (unsigned)(__UINTPTR_TYPE__)(lds_ptr)sbuf;
#pragma unroll
      for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int dblk = 0; dblk < 2; ++dblk) tra[w][dblk] = base + (unsigned)tro[w][dblk];
    }
    auto softmax_tile = [&](f32x16& sc, int j0, float& alpha) {
      float psum;
      mfma_settle(sc);
      if (j0 + KT <= kvlen) {
        const float tmax = pair_max(vmax16(sc));
        float m_new = vmax(m_run, tmax);
        // Deferred rescale (defer_thr > 0): with 64 queries per wave SOME lane's running maximum
        // moves in almost every tile (probability 1 - (1 - 1/t)^64 at tile t), so the wave-uniform
        // branch below rescaled the 32 output registers nearly always.  The reference maximum of a
        // softmax is arbitrary: the wave keeps its old ones -- probabilities up to 2^thr instead
        // of 1, the same RELATIVE bf16 rounding, fp32 sums -- until some lane's maximum grows by
        // more than thr (in log2 units), and then every lane updates, in the textbook order
        // (decide, rescale O and l, exponentiate this tile against the maximum in force).
        if (a.defer_thr > 0.f && !__any((tmax - m_run) * cs > a.defer_thr)) m_new = m_run;
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
          ok[r] = j0 + (r & 3) + 8 * (r >> 2) + 4 * hi < kvlen;
          if (ok[r]) tmax = vmax(tmax, sc[r]);
        }
        tmax = pair_max(tmax);
        const float m_new = vmax(m_run, tmax);
        alpha = __builtin_amdgcn_exp2f((m_run - m_new) * cs);
        const float mc = -m_new * cs;
        psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = ok[r] ? __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], cs, mc)) : 0.f;
          sc[r] = p;
          psum += p;
        }
        m_run = m_new;
      }
      l_run = l_run * alpha + psum;
    };
    // TRA: O^T += V^T P^T of sub-tile (buffer, sb) = idx: the eight transpose reads of both
    // 16-key groups go out together, the first group's MFMAs wait for the older four only
    auto pv_tra = [&](auto idx, const f32x16& sc) {
      typedef short s16x4 __attribute__((ext_vector_type(4)));
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      constexpr int I = decltype(idx)::value;
      constexpr int G0 = (I >> 1) * DSTAGE + DTILE + (I & 1) * 32 * 128, G1 = G0 + 16 * 128;
      s16x4 a0 = tr16_b64_asm<G0>(tra[0][0]), a1 = tr16_b64_asm<G0>(tra[1][0]);
      s16x4 b0 = tr16_b64_asm<G0>(tra[0][1]), b1 = tr16_b64_asm<G0>(tra[1][1]);
      s16x4 c0 = tr16_b64_asm<G1>(tra[0][0]), c1 = tr16_b64_asm<G1>(tra[1][0]);
      s16x4 d0 = tr16_b64_asm<G1>(tra[0][1]), d1 = tr16_b64_asm<G1>(tra[1][1]);
      bf16x8 pa;
#pragma unroll
      for (int e = 0; e < 8; ++e) pa[e] = (__bf16)sc[e];
      asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
          __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7)),
          pa, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
          __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7)),
          pa, o1, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 8; ++e) pa[e] = (__bf16)sc[8 + e];
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c0), "+v"(c1), "+v"(d0), "+v"(d1));
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
          __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(c0, c1, 0, 1, 2, 3, 4, 5, 6, 7)),
          pa, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
          __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(d0, d1, 0, 1, 2, 3, 4, 5, 6, 7)),
          pa, o1, 0, 0, 0);
    };
    auto stage = [&](auto bufc, int it) {
      constexpr int BUF = decltype(bufc)::value;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of stage `it`
      __builtin_amdgcn_s_barrier();                      // everyone's; the other buffer is free
      if (it + 1 < n_it) issue(it + 1, BUF ^ 1);
      const char* sK = sbuf + BUF * DSTAGE;
      const char* sV = sK + DTILE;
      const bool two = it * DKT + KT < kvlen;            // uniform: the stage has a second sub-tile
      bf16x8 kf[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) kf[kk] = *reinterpret_cast<const bf16x8*>(sK + fo[kk]);
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
        if (sb == 1 && !two) break;
        const int j0 = it * DKT + sb * KT;
        f32x16 sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk], qu[kk], sc, 0, 0, 0);
        // under the MFMA chain's tail and the softmax: the next sub-tile's K fragments
        if (sb == 0 && two) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            kf[kk] = *reinterpret_cast<const bf16x8*>(sK + 32 * 128 + fo[kk]);
        }
        __builtin_amdgcn_sched_barrier(0);
        float alpha;
        softmax_tile(sc, j0, alpha);
        if (!__all(alpha == 1.0f)) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            o0[r] *= alpha;
            o1[r] *= alpha;
          }
        }
        if constexpr (TRA) {
          if (sb == 0) pv_tra(std::integral_constant<int, 2 * BUF>{}, sc);
          else pv_tra(std::integral_constant<int, 2 * BUF + 1>{}, sc);
        } else
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bf16x8 pa;
#pragma unroll
          for (int e = 0; e < 8; ++e) pa[e] = (__bf16)sc[8 * j + e];
          bf16x8 v0, v1;
          if constexpr (VTR) {
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            typedef __attribute__((address_space(3))) s16x4* trp;
            const char* vb = sV + (sb * 32 + j * 16) * 128;     // the 16-key group's rows
            const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp)(vb + tro[0][0]));
            const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp)(vb + tro[1][0]));
            const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp)(vb + tro[0][1]));
            const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp)(vb + tro[1][1]));
            v0 = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
            v1 = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
          } else {
            v0 = *reinterpret_cast<const bf16x8*>(sV + fo[2 * sb + j]);
            v1 = *reinterpret_cast<const bf16x8*>(sV + 32 * 128 + fo[2 * sb + j]);
          }
          o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pa, o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pa, o1, 0, 0, 0);
        }
      }
    };
    for (int it = 0; it < n_it; it += 2) {
      stage(std::integral_constant<int, 0>{}, it);
      if (it + 1 < n_it) stage(std::integral_constant<int, 1>{}, it + 1);
    }
  } else
  for (int it = 0; it < n_it; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of stage `it`
    __builtin_amdgcn_s_barrier();                      // everyone's; buffer (it+1)&1 is free
    if (it + 1 < n_it) issue(it + 1, (it + 1) & 1);
    const char* sK = sbuf + (it & 1) * DSTAGE;
    const char* sV = sK + DTILE;
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      const int j0 = it * DKT + sb * KT;
      if (sb == 1 && j0 >= kvlen) break;               // uniform
      // ---- S^T tile ----------------------------------------------------------------
      f32x16 sc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 fk = *reinterpret_cast<const bf16x8*>(
            sK + sb * (32 * 128) + frow + (((2 * kk + hi) ^ sw) << 4));
        sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk, qu[kk], sc, 0, 0, 0);
      }
      // ---- online softmax on this lane's query (see the register-staged kernel) ----------
      float psum, alpha;
      mfma_settle(sc);
      if (j0 + KT <= kvlen) {
        const float tmax = pair_max(vmax16(sc));
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
          ok[r] = j0 + (r & 3) + 8 * (r >> 2) + 4 * hi < kvlen;
          if (ok[r]) tmax = vmax(tmax, sc[r]);
        }
        tmax = pair_max(tmax);
        const float m_new = vmax(m_run, tmax);
        alpha = __builtin_amdgcn_exp2f((m_run - m_new) * cs);
        const float mc = -m_new * cs;
        psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = ok[r] ? __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], cs, mc)) : 0.f;
          sc[r] = p;
          psum += p;
        }
        m_run = m_new;
      }
      l_run = l_run * alpha + psum;
      if (!__all(alpha == 1.0f)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          o0[r] *= alpha;
          o1[r] *= alpha;
        }
      }
      // ---- O^T += V^T P^T ------------------------------------------------------------
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bf16x8 pa;
#pragma unroll
        for (int e = 0; e < 8; ++e) pa[e] = (__bf16)sc[8 * j + e];
        const int vo = frow + (((4 * sb + 2 * j + hi) ^ sw) << 4);
        const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(sV + vo);
        const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(sV + 32 * 128 + vo);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pa, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pa, o1, 0, 0, 0);
      }
    }
  }
  const float l_tot = pair_sum(l_run);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  const int qrow = q0 + wave * 32 + li;
  if (qrow < qlen) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = 8 * g + 4 * hi;
      const f32x4 a0 = f32x4{o0[4 * g], o0[4 * g + 1], o0[4 * g + 2], o0[4 * g + 3]} * inv;
      const f32x4 a1 = f32x4{o1[4 * g], o1[4 * g + 1], o1[4 * g + 2], o1[4 * g + 3]} * inv;
      if (a.o_bf16) {
        __bf16* op = reinterpret_cast<__bf16*>(a.O) + (int64_t)(qoff + qrow) * a.ldo + h * 64;
        bf16x4 b0, b1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { b0[e] = (__bf16)a0[e]; b1[e] = (__bf16)a1[e]; }
        *reinterpret_cast<bf16x4*>(op + d) = b0;
        *reinterpret_cast<bf16x4*>(op + 32 + d) = b1;
      } else {
        float* op = a.O + (int64_t)(qoff + qrow) * a.ldo + h * 64;
        *reinterpret_cast<f32x4*>(op + d) = a0;
        *reinterpret_cast<f32x4*>(op + 32 + d) = a1;
      }
    }
  }
}

template <int NW>
int launch_dma_tra(const AttnArgs& a, hipStream_t s) {
  const int nqb = cdiv(a.max_q_len, NW * 32);
  hipLaunchKernelGGL((attention_bf16_dma_kernel<NW>), dim3(nqb * a.n_heads * a.n_seq),
                     dim3(NW * 64), 0, s, a, nqb);
  WN_HIP(hipGetLastError());
  return 0;
}

template <int NW>
int launch(const AttnArgs& a, hipStream_t s) {
  dim3 g(cdiv(a.max_q_len, NW * 32), a.n_heads, a.n_seq), t(NW * 64);
  if (a.qkv_bf16) {
    if (NW == 8)
      hipLaunchKernelGGL((attention_bf16_kernel<NW, false, 2, true>), g, t, 0, s, a);
    else
      hipLaunchKernelGGL((attention_bf16_kernel<NW, false, 1, true>), g, t, 0, s, a);
  } else if (a.P)
    hipLaunchKernelGGL((attention_bf16_kernel<NW, true>), g, t, 0, s, a);
  else if (NW == 8)
    hipLaunchKernelGGL((attention_bf16_kernel<NW, false, 2, false>), g, t, 0, s, a);
  else
    hipLaunchKernelGGL((attention_bf16_kernel<NW, false>), g, t, 0, s, a);
  WN_HIP(hipGetLastError());
  return 0;
}

}  // namespace


int attention_bf16(const AttnArgs& a, hipStream_t s) {
  // argument checks are attention()'s (the only caller)
  WN_CHECK(a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0 && a.ldp % 4 == 0,
           "attention(bf16): strides must be multiples of 4 elements");
  WN_CHECK(!(a.qkv_bf16 && a.P), "attention(bf16): bf16 Q/K/V only without the rel-pos term");
  int nw = tune().attn_bf16_nw;
  if (nw != 2 && nw != 4 && nw != 8)
    nw = a.max_q_len >= 1024 ? 8 : a.max_q_len >= 384 ? 4 : 2;
  // self attention over bf16 Q | K | V without masks: K and V rows by LDS-DMA, the PV fragments
  // through transpose reads
  if (tune().attn_bf16_dma != 0 && a.qkv_bf16 && !a.P && a.mask_mode == 0 && nw >= 4 &&
      a.q_len == a.kv_len && a.q_off == a.kv_off && a.ldk % 8 == 0 && a.ldv == a.ldk &&
      (int64_t)a.max_q_len * a.ldk * 2 < (int64_t(1) << 31))
  {
    // 4-wave blocks (128 queries) also for long sequences: twice the K / V stream from L2, but
    // barrier groups of four waves lose less to skew than groups of eight (config 5 fp8: 14.81 k
    // vs 14.62 k, r05v); attn_bf16_nw = 8 forces the 256-query blocks
    if (tune().attn_bf16_nw != 8) nw = 4;
    AttnArgs d = a;
    d.defer_thr = 0.1f * (float)tune().attn_bf16_defer;
    return nw == 8 ? launch_dma_tra<8>(d, s) : launch_dma_tra<4>(d, s);
  }
  switch (nw) {
    case 8: return launch<8>(a, s);
    case 4: return launch<4>(a, s);
    default: return launch<2>(a, s);
  }
}

}  // namespace wn
