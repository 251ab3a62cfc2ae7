// HBM-bound / LDS-staged encoder kernels: LayerNorm, CMVN + first subsampling
// conv, the depthwise-conv core of the Conformer conv module, and the
// relative-position multi-head attention (online softmax, fp32 MFMA).
#include "kernels.h"
#include "gemm_epilogue.h"
#include "x6.h"
#include "mxfp8.h"
#include "rowregs.h"

namespace wn {

namespace {

// (RowRegs / ln_inplace: rowregs.h)

// One wave per row; with RPW = 2 a wave owns two adjacent rows and issues both
// row loads before the first reduction (twice the bytes in flight per CU).
// Measured r01f: 3.9 vs 4.1 us isolated on 7932 x 256 (4.2 TB/s, cache
// resident), 11.1 vs 10.8 us on 15800 x 512, no change of the decode step ->
// RPW = 1 stays the default, RPW = 2 is wn_tune_set("ln_rows", 2).
template <int E, int RPW>
__global__ __launch_bounds__(256) void layernorm_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ w,
    const float* __restrict__ b, float* __restrict__ y, int ldy, int M,
    float eps) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
  if (row0 >= M) return;
  RowRegs<E> r[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int row = min(row0 + i, M - 1);
    r[i].load(x + (int64_t)row * ldx, lane);
  }
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    if (row0 + i < M) {
      ln_inplace<E>(r[i], w, b, lane, eps);
      r[i].store(y + (int64_t)(row0 + i) * ldy, lane);
    }
  }
}

// y1 = LN(x; w1, b1); y2 = LN(y1; w2, b2): norm_final of layer i followed by
// norm_ff_macaron of layer i+1 (encoder_layer.py:263 -> :220) in one pass.
template <int E>
__global__ __launch_bounds__(256) void layernorm2_kernel(
    const float* __restrict__ x, const float* __restrict__ w1,
    const float* __restrict__ b1, const float* __restrict__ w2,
    const float* __restrict__ b2, float* __restrict__ y1,
    float* __restrict__ y2, int M, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  constexpr int D = E * 64;
  RowRegs<E> r;
  r.load(x + (int64_t)row * D, lane);
  ln_inplace<E>(r, w1, b1, lane, eps);
  r.store(y1 + (int64_t)row * D, lane);
  ln_inplace<E>(r, w2, b2, lane, eps);
  r.store(y2 + (int64_t)row * D, lane);
}

// bf16-storage form of the bf16 mode: the normalised row only ever feeds a GEMM,
// whose first step rounds it to bf16 anyway, so it is WRITTEN as bf16 (half the
// bytes out, half the bytes into the GEMM, identical arithmetic).
template <int E>
__device__ __forceinline__ void store_row_bf16(const RowRegs<E>& r, __bf16* p, int lane) {
  constexpr int VEC = RowRegs<E>::VEC;
  typedef __bf16 hvec_t __attribute__((ext_vector_type(VEC)));
#pragma unroll
  for (int j = 0; j < E / VEC; ++j) {
    if constexpr (VEC == 1) {
      p[j * 64 + lane] = (__bf16)r.v[j];
    } else {
      hvec_t t;
#pragma unroll
      for (int e = 0; e < VEC; ++e) t[e] = (__bf16)r.v[j * VEC + e];
      *reinterpret_cast<hvec_t*>(p + j * 64 * VEC + lane * VEC) = t;
    }
  }
}

template <int E>
__global__ __launch_bounds__(256) void layernorm_bf16out_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ w,
    const float* __restrict__ b, __bf16* __restrict__ y, int ldy, int M, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  RowRegs<E> r;
  r.load(x + (int64_t)row * ldx, lane);
  ln_inplace<E>(r, w, b, lane, eps);
  store_row_bf16<E>(r, y + (int64_t)row * ldy, lane);
}

// MXFP8 form of the fp8 mode (WN_PREC_FP8): the normalised row feeds the FFN w_1 GEMM
// as e4m3 elements + one E8M0 scale per 32 columns (csrc/mxfp8.h).  A lane holds 4
// consecutive columns per 256-column chunk, 8 lanes one MX block.
template <int E>
__global__ __launch_bounds__(256) void layernorm_mx_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ w,
    const float* __restrict__ b, unsigned char* __restrict__ q, unsigned* __restrict__ scale,
    int pitch, int M, float eps) {
  static_assert(E % 4 == 0, "MX LayerNorm: 4 columns per lane and chunk");
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  constexpr int D = E * 64;
  RowRegs<E> r;
  r.load(x + (int64_t)row * ldx, lane);
  ln_inplace<E>(r, w, b, lane, eps);
#pragma unroll
  for (int j = 0; j < E / 4; ++j) {
    const float* v = r.v + 4 * j;
    float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
    const int Eb = mx_e8m0(amax);
    const float inv = mx_inv_scale(Eb);
    const int c = j * 256 + lane * 4;
    *reinterpret_cast<int*>(q + (int64_t)row * D + c) =
        mx_pack4(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv);
    if ((lane & 7) == 0)
      reinterpret_cast<unsigned char*>(scale)[((int64_t)(c >> 7) * pitch + row) * 4 +
                                              ((c >> 5) & 3)] = (unsigned char)Eb;
  }
}

template <int E>
__global__ __launch_bounds__(256) void layernorm2_bf16out_kernel(
    const float* __restrict__ x, const float* __restrict__ w1,
    const float* __restrict__ b1, const float* __restrict__ w2,
    const float* __restrict__ b2, float* __restrict__ y1, __bf16* __restrict__ y2,
    int M, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  constexpr int D = E * 64;
  RowRegs<E> r;
  r.load(x + (int64_t)row * D, lane);
  ln_inplace<E>(r, w1, b1, lane, eps);
  r.store(y1 + (int64_t)row * D, lane);
  ln_inplace<E>(r, w2, b2, lane, eps);
  store_row_bf16<E>(r, y2 + (int64_t)row * D, lane);
}

// ===========================================================================
// GlobalCMVN (cmvn.py:36-47) + Conv2d(1, C, 3, stride 2) + ReLU
// (subsampling.py:188-190).  One block per (utterance, T1 frame): the three
// normalised input rows sit in LDS, threads sweep (f1, c) with c fastest so
// the channels-last output row is written in full 16-byte segments.
__global__ __launch_bounds__(256) void cmvn_conv1_kernel(Conv1Args a) {
  const int b = blockIdx.y;
  const int t1 = blockIdx.x;
  if (t1 >= a.t1_len[b]) return;
  __shared__ float xin[3][128];
  const float* src = a.feats + ((int64_t)b * a.T + 2 * t1) * a.F;
  for (int i = threadIdx.x; i < 3 * a.F; i += 256) {
    const int r = i / a.F, f = i % a.F;
    float v = src[r * a.F + f];
    if (a.mean) v = (v - a.mean[f]) * a.istd[f];
    xin[r][f] = v;
  }
  __syncthreads();
  float* dst = a.out + (int64_t)(a.t1_off[b] + t1) * a.F1 * a.C;
  // one output channel per thread: its 9 taps + bias stay in registers, the
  // input taps are wave-uniform LDS broadcasts, the store is one coalesced
  // row of C floats per f1
  for (int c = threadIdx.x; c < a.C; c += 256) {
    float w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = a.w[k * a.C + c];
    const float bias = a.bias[c];
    for (int f1 = 0; f1 < a.F1; ++f1) {
      // accumulation order = (ky, kx) row-major, like a direct conv loop
      float acc = bias;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
          acc = fmaf(w[ky * 3 + kx], xin[ky][2 * f1 + kx], acc);
      dst[f1 * a.C + c] = fmaxf(acc, 0.f);
    }
  }
}

// The same convolution written as the plane image of its output (gemm_x6.hip) for the
// six-product conv2: one image row per conv1 pixel, pixel index = frame * F1 + pos with the
// even f1 first (pos = f1 / 2) and the odd ones behind them (pos = (F1 + 1) / 2 + f1 / 2),
// so that the stride-2 taps of conv2 read CONSECUTIVE pixels.  A wave owns ONE group of 8
// channels (its 72 weights + 8 biases are loaded once, wave-uniform) and walks the pixels
// of CF1 consecutive frames, 64 per pass (lane = pixel): a plane store covers consecutive
// 16-B pieces.  Same accumulation order as above and an exact split -- conv2 sees the same
// operand values.  (First version: one frame per block, the channel groups looped inside:
// 375 us at config 2, waiting for the scalar weight loads of every iteration and with 39
// of 64 lanes busy; the 976-MB image alone is ~200 us of HBM writes.)
constexpr int CF1 = 16;    // frames per block
__global__ __launch_bounds__(256) void cmvn_conv1_x3_kernel(Conv1Args a) {
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * CF1;
  const int nf = min(CF1, a.t1_len[b] - f0);
  if (nf <= 0) return;
  __shared__ float xin[2 * CF1 + 1][128];
  const float* src = a.feats + ((int64_t)b * a.T + 2 * f0) * a.F;
  for (int i = threadIdx.x; i < (2 * nf + 1) * a.F; i += 256) {
    const int r = i / a.F, f = i - r * a.F;
    float v = src[i];
    if (a.mean) v = (v - a.mean[f]) * a.istd[f];
    xin[r][f] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cg = blockIdx.z * 4 + wave;            // channels cg*8 .. +8
  float w[9][8], bias[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias[e] = a.bias[cg * 8 + e];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) w[k][e] = a.w[k * a.C + cg * 8 + e];
  const int ne = (a.F1 + 1) / 2;
  const int total = nf * a.F1;
  const int64_t P0 = (int64_t)(a.t1_off[b] + f0) * a.F1;
  for (int px = lane; px < total; px += 64) {
    const int fr = px / a.F1, pos = px - fr * a.F1;
    const int f1 = pos < ne ? 2 * pos : 2 * (pos - ne) + 1;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = bias[e];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float x = xin[2 * fr + ky][2 * f1 + kx];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(w[ky * 3 + kx][e], x, acc[e]);
      }
    bf16x8 p0, p1, p2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const Split3 sp = split3(fmaxf(acc[e], 0.f));
      p0[e] = sp.h0; p1[e] = sp.h1; p2[e] = sp.h2;
    }
    char* o = a.out3 + x3_piece(cg >> 1, a.tiles, (int)(P0 + px), cg & 1);
    *reinterpret_cast<bf16x8*>(o) = p0;
    *reinterpret_cast<bf16x8*>(o + X3_REC) = p1;
    *reinterpret_cast<bf16x8*>(o + 2 * X3_REC) = p2;
  }
}

// ===========================================================================
// Depthwise Conv1d over time + LayerNorm(channels) + SiLU, the middle of
// ConvolutionModule.forward (convolution.py:119-146).  One wave per output
// frame, channels in registers.  A neighbour frame outside the utterance is
// what the reference would have read there:
//   t < 0           : causal -> GLU(pointwise_conv1(0)) = cpad (the K-1 left
//                     pad happens BEFORE pointwise_conv1, :122-124);
//                     symmetric -> 0 (Conv1d zero padding)
//   len <= t < Tmax : cpad (masked_fill(~mask_pad, 0) before pointwise_conv1)
//   t >= Tmax       : 0 (symmetric Conv1d zero padding past the tensor end)
template <int E>
__global__ __launch_bounds__(256) void dwconv_kernel(DwConvArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.M) return;
  const int u = a.row_utt[row];
  if (u < 0) return;
  const int off = a.off[u], len = a.len[u];
  const int t = row - off;
  if (t >= len) return;
  const int lpad = a.causal ? a.K - 1 : (a.K - 1) / 2;
  RowRegs<E> acc, cp;
  acc.load(a.bias, lane);
  cp.load(a.cpad, lane);
  // taps in groups of 8: ALL loads of a group (taps + the 8 neighbour rows) are issued before
  // the first fma -- one memory round trip per group instead of one per tap (13.9 -> 13.1 us at
  // M = 7932, K = 8, r03t; the order of the fmas is unchanged)
  constexpr int TG = E <= 8 ? 8 : 4;
  for (int k0 = 0; k0 < a.K; k0 += TG) {
    RowRegs<E> wk[TG], xv[TG];
    int kind[TG];          // 0: outside (nothing), 1: row of the utterance, 2: the pad frame
#pragma unroll
    for (int i = 0; i < TG; ++i) {
      const int k = k0 + i, tt = t + k - lpad;
      kind[i] = 0;
      if (k < a.K) {
        wk[i].load(a.wt + (int64_t)k * (E * 64), lane);
        if (tt >= 0 && tt < len) {
          kind[i] = 1;
          xv[i].load(a.x + (int64_t)(off + tt) * a.ldx, lane);
        } else if ((tt < 0 && a.causal) || (tt >= len && tt < a.t_max)) {
          kind[i] = 2;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < TG; ++i) {
      if (kind[i] == 1) {
#pragma unroll
        for (int e = 0; e < E; ++e) acc.v[e] = fmaf(wk[i].v[e], xv[i].v[e], acc.v[e]);
      } else if (kind[i] == 2) {
#pragma unroll
        for (int e = 0; e < E; ++e) acc.v[e] = fmaf(wk[i].v[e], cp.v[e], acc.v[e]);
      }
    }
  }
  if (a.norm_mode == 0) {
    ln_inplace<E>(acc, a.ln_w, a.ln_b, lane, a.eps);
  } else {
    RowRegs<E> sc, sh;
    sc.load(a.ln_w, lane);
    sh.load(a.ln_b, lane);
#pragma unroll
    for (int e = 0; e < E; ++e) acc.v[e] = fmaf(acc.v[e], sc.v[e], sh.v[e]);
  }
#pragma unroll
  for (int e = 0; e < E; ++e) acc.v[e] = silu_f(acc.v[e]);
  acc.store(a.y + (int64_t)row * a.ldy, lane);
}

// dwconv_tiled = 1 (the default since round 4; bit-identical to the kernel above): a wave
// computes R = 4 consecutive packed rows.  The kernel above reads K neighbour rows per output
// row -- 8 x 1 KB from L2 per row at config 2, 65 MB per launch for an 8-MB tensor; a tile of 4
// rows shares its R + 7 window rows per group of 8 taps (11 row loads instead of 32, the taps
// once instead of four times).  Per output row the operations and their order are the kernel
// above's: bias, the taps in ascending order (a neighbour outside the utterance is the pad
// frame or skipped by the same rule, evaluated against the OUTPUT row's own utterance -- a tile
// may straddle two utterances of the packed batch), LayerNorm / affine, SiLU.
template <int E>
__global__ __launch_bounds__(256) void dwconv_tiled_kernel(DwConvArgs a) {
  constexpr int R = 4, TG = 8, NWIN = R + TG - 1;
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= a.M) return;
  const int lpad = a.causal ? a.K - 1 : (a.K - 1) / 2;
  // utterance, first row and length of the R rows: lanes 0 .. R - 1 fetch them side by side (two
  // dependent round trips for the tile, as for one row above), the wave reads them back as
  // uniform values
  int u_l = -1, off_l = 0, len_l = 0;
  if (lane < R && row0 + lane < a.M) u_l = a.row_utt[row0 + lane];
  if (u_l >= 0) {
    off_l = a.off[u_l];
    len_l = a.len[u_l];
  }
  int t_r[R], len_r[R];
  bool on[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int u = __builtin_amdgcn_readlane(u_l, r);
    t_r[r] = row0 + r - __builtin_amdgcn_readlane(off_l, r);
    len_r[r] = __builtin_amdgcn_readlane(len_l, r);
    on[r] = u >= 0 && t_r[r] < len_r[r];
  }
  RowRegs<E> acc[R], cp;
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r].load(a.bias, lane);
  cp.load(a.cpad, lane);
  for (int k0 = 0; k0 < a.K; k0 += TG) {
    RowRegs<E> wk[TG], xw[NWIN];
    // window row i = packed row row0 + k0 - lpad + i (output r, tap k0 + j: i = r + j); the
    // address is clamped into the tensor, whether the row is USED is decided per output below
#pragma unroll
    for (int i = 0; i < NWIN; ++i) {
      const int p = min(max(row0 + k0 - lpad + i, 0), a.M - 1);
      xw[i].load(a.x + (int64_t)p * a.ldx, lane);
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
    if (!on[r]) continue;
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
    acc[r].store(a.y + (int64_t)(row0 + r) * a.ldy, lane);
  }
}

// ===========================================================================
// Attention.  Replaces RelPositionMultiHeadedAttention.forward /
// MultiHeadedAttention.forward_attention (attention.py:133-178,364-438) and
// the decoder's self / cross attention.  No (T x T) score tensor ever reaches
// HBM.
//
// One block = NW waves, each wave owns 32 query rows of one (sequence, head).
// Per 32-key tile, K, V (and the projected position rows P) are staged
// global -> registers -> LDS one tile ahead (double-buffered LDS, one barrier
// per tile: the load latency hides under the previous tile's MFMAs) and
// the TRANSPOSED score tile S^T = K (Q+u)^T + P (Q+v)^T is produced with
// v_mfma_f32_32x32x2_f32: lane l then holds, for ITS query (l & 31), the 16
// keys (r&3)+8(r>>2)+4(l>>5), r = 0..15.  Row max / sum are 15 in-lane ops plus
// one exchange with lane^32, and -- the point of the transposed form -- the
// probabilities are already in the A-operand layout of the P.V MFMA: step s
// consumes register p[s] directly against the V rows key(s, lane>>5).
constexpr int KT = 32;          // keys per tile
constexpr int KSTR = 68;        // LDS row stride (floats): 64 + 4 pad

// KS = 2 ("key split"): the block has NW * 2 waves; waves [0, NW) take the first
// half of the block's key tiles, waves [NW, 2 NW) the second half, for the SAME
// NW * 32 queries, and the two partial (max, sum, O) states are merged through
// LDS at the end.  Twice the waves per SIMD for the same staging traffic: with
// one 32-query task per SIMD (config 2: 1062 tasks on 1024 SIMDs) every LDS /
// MFMA latency of the dependent chain was exposed.  The LDS budget stays 52 KB
// per block: one buffer per half instead of two buffers (two barriers per
// iteration; the other resident waves cover them).
// FOLD (with RELPOS false): the rel-pos term folded into the keys while they are staged --
// see relpos_fold_kernel below for the algebra: the thread that stages a float4 of key row j
// adds the same float4 of position row j (k <- k + p) and forms its share of u.k + v.p; 16
// consecutive threads hold one key row, so four shuffles complete the scalar, which goes to
// LDS next to the tile and is added to the score before the scale.
// GLB (the folded key-split kernel; the default since round 4, bit-identical to the chunk-by-chunk
// form it replaced): the K / V / P loads of all of a thread's chunks are issued before the first
// fold.  Chunk by chunk, every chunk's three loads were followed by their own s_waitcnt and the
// chunk's dot product + four shuffles before the next chunk's loads went out (read off the ISA at
// the end of round 3): NCH dependent memory round trips per staging step instead of one; same
// arithmetic per element.  (KS == 1 keeps the chunk-by-chunk order: its loads are a register
// prefetch one tile ahead.)
template <int NW, bool RELPOS, int KS, bool FOLD = false>
__global__ __launch_bounds__(NW * KS * 64, KS == 2 ? 3 : 1) void attention_kernel(AttnArgs a) {
  constexpr bool GLB = FOLD && KS == 2;
  int s = blockIdx.z, h = blockIdx.y, qb = blockIdx.x;
  if (a.blk_tab) {       // the launch's block list (kernels.h AttnArgs::blk_tab)
    const int e = a.blk_tab[blockIdx.x];
    s = e >> 16; h = (e >> 8) & 255; qb = e & 255;
  } else if (a.xcd_nqb > 0) {   // XCD-aware block order (kernels.h AttnArgs::xcd_nqb)
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
  const int p_off = a.p_off ? a.p_off[s] : 0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably uniform
  const int wave = wave_all % NW;   // query group
  const int kh = wave_all / NW;     // key half (0 when KS == 1)
  const int hi = lane >> 5, li = lane & 31;
  constexpr int NTHR = NW * KS * 64;

  // K / V / P tiles: KS == 1: [2 buffers][matrix][KT * KSTR], double buffered;
  //                  KS == 2: [2 halves][matrix][KT * KSTR], single buffered
  constexpr int NMAT = RELPOS ? 3 : 2;
  constexpr int MAT = KT * KSTR;
  __shared__ __attribute__((aligned(16))) float stile[2 * NMAT * MAT];
  // per-key additive score term of the folded rel-pos form (a.kbias): one float per key of
  // the tile, staged with it ([buffer | half][key])
  __shared__ float sbias[2][KT];
  const bool kb_on = FOLD || (!RELPOS && a.kbias != nullptr);

  // ---- this lane's query row ---------------------------------------------
  const int qi = q0 + wave * 32 + li;
  const int qc = qi < qlen ? qi : qlen - 1;
  f32x4 qu[8], qv[8];
  {
    const float* qp = a.Q + (int64_t)(qoff + qc) * a.ldq + h * 64 + hi * 4;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      f32x4 q = *reinterpret_cast<const f32x4*>(qp + kk * 8);
      if (RELPOS) {
        const f32x4 bu = *reinterpret_cast<const f32x4*>(
            a.bias_u + h * 64 + kk * 8 + hi * 4);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(
            a.bias_v + h * 64 + kk * 8 + hi * 4);
        qu[kk] = q + bu;
        qv[kk] = q + bv;
      } else {
        qu[kk] = q;
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
  // iterations of the tile loop; half kh works on tile t_lo + kh * n_it + it
  const int n_it = KS == 1 ? (t_hi - t_lo) : (t_hi - t_lo + 1) / 2;

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;
  // (a query group past the sequence's end only stages and meets the barriers: attention_x6.hip)
  const bool wave_live = q0 + wave * 32 < qlen;

  // ---- tile staging: global -> registers (issued a whole iteration ahead, so
  // the HBM / L2 latency hides under the MFMAs) -> LDS.  With KS == 2 one
  // staging step moves the tiles of BOTH halves.
  constexpr int NCH = KT * 16 * KS / NTHR;  // float4 chunks per thread and matrix
  f32x4 rK[NCH], rV[NCH], rP[RELPOS ? NCH : 1];
  float rC = 0.f, rCf[FOLD ? NCH : 1];
  f32x4 fu = {0.f, 0.f, 0.f, 0.f}, fv = fu;
  if (FOLD) {   // NTHR is a multiple of 16: a thread's column quad (tid & 15) never changes
    fu = *reinterpret_cast<const f32x4*>(a.bias_u + h * 64 + (tid & 15) * 4);
    fv = *reinterpret_cast<const f32x4*>(a.bias_v + h * 64 + (tid & 15) * 4);
  }
  auto gload = [&](int it) {
    if constexpr (GLB) {
      f32x4 rPf[NCH];
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = tid + i * NTHR;
        const int half = KS == 1 ? 0 : c / (KT * 16);
        const int w = c % (KT * 16);
        const int r = w >> 4, c4 = w & 15;
        int j = (t_lo + half * n_it + it) * KT + r;
        if (j > kvlen - 1) j = kvlen - 1;
        const int64_t grow = kvoff + j;
        rK[i] = *reinterpret_cast<const f32x4*>(a.K + grow * a.ldk + h * 64 + c4 * 4);
        rV[i] = *reinterpret_cast<const f32x4*>(a.V + grow * a.ldv + h * 64 + c4 * 4);
        rPf[i] = *reinterpret_cast<const f32x4*>(a.P + (int64_t)(j + p_off) * a.ldp + h * 64 +
                                                 c4 * 4);
      }
      // (keeps the loads above in front of the first fold)
#pragma unroll
      for (int i = 0; i < NCH; ++i) asm volatile("" : "+v"(rK[i]), "+v"(rPf[i]));
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const f32x4 p = rPf[i];
        const f32x4 k = rK[i];
        float d = fu[0] * k[0] + fu[1] * k[1] + fu[2] * k[2] + fu[3] * k[3] +
                  fv[0] * p[0] + fv[1] * p[1] + fv[2] * p[2] + fv[3] * p[3];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
        rCf[i] = d;
        rK[i] = k + p;
      }
      return;
    }
    if (!FOLD && kb_on && tid < KT * KS) {
      int j = (t_lo + (tid / KT) * n_it + it) * KT + (tid % KT);
      if (j > kvlen - 1) j = kvlen - 1;
      rC = a.kbias[(int64_t)(kvoff + j) * a.n_heads + h];
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * NTHR;
      const int half = KS == 1 ? 0 : c / (KT * 16);
      const int w = c % (KT * 16);
      const int r = w >> 4, c4 = w & 15;
      int j = (t_lo + half * n_it + it) * KT + r;
      if (j > kvlen - 1) j = kvlen - 1;
      const int64_t grow = kvoff + j;
      rK[i] = *reinterpret_cast<const f32x4*>(a.K + grow * a.ldk + h * 64 + c4 * 4);
      rV[i] = *reinterpret_cast<const f32x4*>(a.V + grow * a.ldv + h * 64 + c4 * 4);
      if (RELPOS)
        rP[i] = *reinterpret_cast<const f32x4*>(a.P + (int64_t)(j + p_off) * a.ldp +
                                                h * 64 + c4 * 4);
      if (FOLD) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(a.P + (int64_t)(j + p_off) * a.ldp +
                                                        h * 64 + c4 * 4);
        const f32x4 k = rK[i];
        float d = fu[0] * k[0] + fu[1] * k[1] + fu[2] * k[2] + fu[3] * k[3] +
                  fv[0] * p[0] + fv[1] * p[1] + fv[2] * p[2] + fv[3] * p[3];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
        rCf[i] = d;
        rK[i] = k + p;
      }
    }
  };
  auto lstore = [&](int buf) {  // KS == 1: buffer index; KS == 2: ignored
    if (!FOLD && kb_on && tid < KT * KS) sbias[KS == 1 ? buf : tid / KT][tid % KT] = rC;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * NTHR;
      const int half = KS == 1 ? buf : c / (KT * 16);
      const int w = c % (KT * 16);
      const int r = w >> 4, c4 = w & 15;
      if (FOLD && c4 == 0) sbias[half][r] = rCf[i];
      float* base = stile + half * NMAT * MAT;
      *reinterpret_cast<f32x4*>(base + r * KSTR + c4 * 4) = rK[i];
      *reinterpret_cast<f32x4*>(base + MAT + r * KSTR + c4 * 4) = rV[i];
      if (RELPOS)
        *reinterpret_cast<f32x4*>(base + 2 * MAT + r * KSTR + c4 * 4) = rP[i];
    }
  };
  if (KS == 1 && n_it > 0) {
    gload(0);
    lstore(0);
  }
  if (KS == 1) __syncthreads();

  for (int it = 0; it < n_it; ++it) {
    const int kt = t_lo + kh * n_it + it;   // this wave's tile (may be >= t_hi)
    const int j0 = kt * KT;
    const int cur = KS == 1 ? (it & 1) : kh;
    const float* sK = stile + cur * NMAT * MAT;
    const float* sV = sK + MAT;
    const float* sP = sK + 2 * MAT;
    if (KS == 2) {
      // no register prefetch here: three waves per SIMD (<= 168 VGPRs, 3 blocks
      // of 52 KB per CU) cover the load latency, and 768 block slots take the
      // ~531 blocks of config 2 in ONE round
      gload(it);
      lstore(0);
      __syncthreads();  // both halves' tiles visible
    } else if (it + 1 < n_it) {
      gload(it + 1);
    }

    if ((KS == 1 || kt < t_hi) && wave_live) {
    // ---- S^T tile -------------------------------------------------------------
    f32x16 sc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.f;
    const float* kf = sK + li * KSTR + hi * 4;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const f32x4 fk = *reinterpret_cast<const f32x4*>(kf + kk * 8);
#pragma unroll
      for (int t = 0; t < 4; ++t)
        sc = __builtin_amdgcn_mfma_f32_32x32x2f32(fk[t], qu[kk][t], sc, 0, 0, 0);
    }
    if (RELPOS) {
      const float* pf = sP + li * KSTR + hi * 4;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const f32x4 fp = *reinterpret_cast<const f32x4*>(pf + kk * 8);
#pragma unroll
        for (int t = 0; t < 4; ++t)
          sc = __builtin_amdgcn_mfma_f32_32x32x2f32(fp[t], qv[kk][t], sc, 0, 0,
                                                    0);
      }
    }
    // ---- online softmax on this lane's query -----------------------------------
    float tmax = -1e30f;
    bool ok[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      ok[r] = (j >= jmin) && (j < jmax);
      if (kb_on) sc[r] += sbias[cur][(r & 3) + 8 * (r >> 2) + 4 * hi];
      sc[r] *= a.scale;
      if (ok[r]) tmax = fmaxf(tmax, sc[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = ok[r] ? __expf(sc[r] - m_new) : 0.f;
      sc[r] = p;
      psum += p;
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    // rescale the running output: its rows are queries (r&3)+8(r>>2)+4hi
    if (!__all(alpha == 1.0f)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float ar = __shfl(alpha, (r & 3) + 8 * (r >> 2) + 4 * hi, 64);
        o0[r] *= ar;
        o1[r] *= ar;
      }
    }
    // ---- O += P V ----------------------------------------------------------------
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      const int key = (st & 3) + 8 * (st >> 2) + 4 * hi;
      const float v0 = sV[key * KSTR + li];
      const float v1 = sV[key * KSTR + 32 + li];
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(sc[st], v0, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(sc[st], v1, o1, 0, 0, 0);
    }
    }
    if (KS == 1 && it + 1 < n_it) lstore(cur ^ 1);
    __syncthreads();  // KS == 1: next tile visible, this buffer free;
                      // KS == 2: every wave is done with the tiles
  }
  float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (KS == 2) {
    // ---- merge the two key halves: half 1 parks (m, l, O) in LDS, half 0
    // folds it in.  O rows are queries (r&3)+8(r>>2)+4hi, columns li / 32+li.
    float* xm = stile + wave * (32 * 65 + 64);   // per query group: m[32] l[32] O[32][65]
    float* xo = xm + 64;
    if (kh == 1) {
      if (hi == 0) { xm[li] = m_run; xm[32 + li] = l_tot; }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qr = (r & 3) + 8 * (r >> 2) + 4 * hi;
        xo[qr * 65 + li] = o0[r];
        xo[qr * 65 + 32 + li] = o1[r];
      }
    }
    __syncthreads();
    if (kh == 1) return;
    const float m1 = xm[li], l1 = xm[32 + li];
    const float m = fmaxf(m_run, m1);
    const float a0 = __expf(m_run - m), a1 = __expf(m1 - m);
    l_tot = l_tot * a0 + l1 * a1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qr = (r & 3) + 8 * (r >> 2) + 4 * hi;
      const float b0 = __shfl(a0, qr, 64), b1 = __shfl(a1, qr, 64);
      o0[r] = o0[r] * b0 + xo[qr * 65 + li] * b1;
      o1[r] = o1[r] * b0 + xo[qr * 65 + 32 + li] * b1;
    }
  }
  // ---- normalise and store -------------------------------------------------
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;  // fully-masked row -> 0
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int qr = (r & 3) + 8 * (r >> 2) + 4 * hi;
    const float ir = __shfl(inv, qr, 64);
    const int qrow = q0 + wave * 32 + qr;
    if (qrow < qlen) {
      float* op = a.O + (int64_t)(qoff + qrow) * a.ldo + h * 64;
      op[li] = o0[r] * ir;
      op[32 + li] = o1[r] * ir;
    }
  }
}

// Rel-pos attention folded into plain attention.  RelPositionMultiHeadedAttention scores
// (attention.py:410-428, this version without rel_shift: position row j for key j)
//     (q + u) . k_j + (q + v) . p_j  =  q . (k_j + p_j)  +  (u . k_j + v . p_j)
// -- ONE contraction with the keys k'_j = k_j + p_j and a per-key, per-head scalar instead
// of two contractions per score (a third of the kernel's MFMA work and of its LDS traffic:
// the P tile is gone).  This kernel rewrites K in place and writes the scalars; one wave
// per key row, 4 columns per lane and 256-column chunk (head = column / 64).
template <int E>
__global__ __launch_bounds__(256) void relpos_fold_kernel(
    float* __restrict__ K, int ldk, const float* __restrict__ P, int ldp,
    const float* __restrict__ bias_u, const float* __restrict__ bias_v,
    const int* __restrict__ row_utt, const int* __restrict__ off,
    const int* __restrict__ p_off, float* __restrict__ kbias, int n_heads, int M) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int u = row_utt[row];
  if (u < 0) return;
  const int j = row - off[u] + (p_off ? p_off[u] : 0);
#pragma unroll
  for (int c = 0; c < E / 4; ++c) {
    const int col = c * 256 + lane * 4;
    float* kp = K + (int64_t)row * ldk + col;
    const f32x4 k = *reinterpret_cast<const f32x4*>(kp);
    const f32x4 p = *reinterpret_cast<const f32x4*>(P + (int64_t)j * ldp + col);
    const f32x4 bu = *reinterpret_cast<const f32x4*>(bias_u + col);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_v + col);
    float d = bu[0] * k[0] + bu[1] * k[1] + bu[2] * k[2] + bu[3] * k[3] +
              bv[0] * p[0] + bv[1] * p[1] + bv[2] * p[2] + bv[3] * p[3];
    // a head = 64 columns = 16 lanes
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    *reinterpret_cast<f32x4*>(kp) = k + p;
    if ((lane & 15) == 0) kbias[(int64_t)row * n_heads + c * 4 + (lane >> 4)] = d;
  }
}

__global__ void copy_rows_kernel(const float* src, int lds, const int* src_rows,
                                 float* dst, int ldd, const int* dst_rows,
                                 int n_rows, int D4) {
  const int r = blockIdx.x;
  if (r >= n_rows) return;
  const int sr = src_rows ? src_rows[r] : r;
  const int dr = dst_rows ? dst_rows[r] : r;
  if (sr < 0 || dr < 0) return;
  const f32x4* s = reinterpret_cast<const f32x4*>(src + (int64_t)sr * lds);
  f32x4* d = reinterpret_cast<f32x4*>(dst + (int64_t)dr * ldd);
  for (int i = threadIdx.x; i < D4; i += blockDim.x) d[i] = s[i];
}

}  // namespace

int layernorm_mx(const float* x, int ldx, const float* w, const float* b, void* q,
                 unsigned* scale, int pitch, int M, int D, float eps, hipStream_t s) {
  WN_CHECK(M > 0 && ldx % 4 == 0 && D % 256 == 0 && pitch >= M,
           "layernorm_mx: width must be a multiple of 256");
  dim3 g(cdiv(M, 4)), t(256);
  unsigned char* qq = reinterpret_cast<unsigned char*>(q);
#define WN_LNM(E)                                                                   \
  case E * 64:                                                                      \
    hipLaunchKernelGGL((layernorm_mx_kernel<E>), g, t, 0, s, x, ldx, w, b, qq, scale, \
                       pitch, M, eps);                                              \
    break;
  switch (D) {
    WN_LNM(4) WN_LNM(8) WN_LNM(12) WN_LNM(16) WN_LNM(20)
    default:
      set_error("layernorm_mx: unsupported width " + std::to_string(D));
      return -1;
  }
#undef WN_LNM
  WN_HIP(hipGetLastError());
  return 0;
}

int layernorm(const float* x, int ldx, const float* w, const float* b, float* y,
              int ldy, int M, int D, float eps, hipStream_t s, bool y_bf16) {
  WN_CHECK(M > 0, "layernorm: empty");
  WN_CHECK(ldx % 4 == 0 && ldy % 4 == 0, "layernorm: row stride % 4");
  if (y_bf16) {  // y is a bf16 matrix, ldy in bf16 elements
    dim3 gh(cdiv(M, 4)), th(256);
    __bf16* yh = reinterpret_cast<__bf16*>(y);
#define WN_LNH(E)                                                                  \
  case E * 64:                                                                    \
    hipLaunchKernelGGL((layernorm_bf16out_kernel<E>), gh, th, 0, s, x, ldx, w, b,  \
                       yh, ldy, M, eps);                                          \
    break;
    switch (D) {
      WN_LNH(1) WN_LNH(2) WN_LNH(3) WN_LNH(4) WN_LNH(6) WN_LNH(8) WN_LNH(10)
      WN_LNH(12) WN_LNH(16) WN_LNH(20)
      default:
        set_error("layernorm: unsupported width " + std::to_string(D));
        return -1;
    }
#undef WN_LNH
    WN_HIP(hipGetLastError());
    return 0;
  }
  // (two rows per wave measured no gain, docs/LOG_rounds1-3.md section 6: one row per wave only)
  dim3 g(cdiv(M, 4)), t(256);
#define WN_LN(E)                                                                 \
  case E * 64:                                                                   \
    hipLaunchKernelGGL((layernorm_kernel<E, 1>), g, t, 0, s, x, ldx, w, b, y,    \
                       ldy, M, eps);                                             \
    break;
  switch (D) {
    WN_LN(1) WN_LN(2) WN_LN(3) WN_LN(4) WN_LN(6) WN_LN(8) WN_LN(10) WN_LN(12)
    WN_LN(16) WN_LN(20)
    default:
      set_error("layernorm: unsupported width " + std::to_string(D));
      return -1;
  }
#undef WN_LN
  WN_HIP(hipGetLastError());
  return 0;
}

int layernorm2(const float* x, const float* w1, const float* b1, const float* w2,
               const float* b2, float* y1, float* y2, int M, int D, float eps,
               hipStream_t s, bool y2_bf16) {
  WN_CHECK(M > 0, "layernorm2: empty");
  dim3 g(cdiv(M, 4)), t(256);
  if (y2_bf16) {
    __bf16* yh = reinterpret_cast<__bf16*>(y2);
#define WN_LN2H(E)                                                                 \
  case E * 64:                                                                    \
    hipLaunchKernelGGL(layernorm2_bf16out_kernel<E>, g, t, 0, s, x, w1, b1, w2,   \
                       b2, y1, yh, M, eps);                                       \
    break;
    switch (D) {
      WN_LN2H(1) WN_LN2H(2) WN_LN2H(4) WN_LN2H(8) WN_LN2H(12) WN_LN2H(16) WN_LN2H(20)
      default:
        set_error("layernorm2: unsupported width " + std::to_string(D));
        return -1;
    }
#undef WN_LN2H
    WN_HIP(hipGetLastError());
    return 0;
  }
#define WN_LN2(E)                                                               \
  case E * 64:                                                                  \
    hipLaunchKernelGGL(layernorm2_kernel<E>, g, t, 0, s, x, w1, b1, w2, b2, y1, \
                       y2, M, eps);                                             \
    break;
  switch (D) {
    WN_LN2(1) WN_LN2(2) WN_LN2(4) WN_LN2(8) WN_LN2(12) WN_LN2(16) WN_LN2(20)
    default:
      set_error("layernorm2: unsupported width " + std::to_string(D));
      return -1;
  }
#undef WN_LN2
  WN_HIP(hipGetLastError());
  return 0;
}

int cmvn_conv1_relu(const Conv1Args& a, hipStream_t s) {
  WN_CHECK(a.F <= 128, "conv1: feature dim > 128");
  WN_CHECK(a.max_t1 > 0 && a.B > 0, "conv1: empty");
  if (a.out3) {
    WN_CHECK(a.F1 <= 64 && a.C % 32 == 0 && a.tiles > 0, "conv1: plane image shape");
    hipLaunchKernelGGL(cmvn_conv1_x3_kernel, dim3(cdiv(a.max_t1, CF1), a.B, a.C / 32), dim3(256),
                       0, s, a);
    WN_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(cmvn_conv1_kernel, dim3(a.max_t1, a.B), dim3(256), 0, s,
                     a);
  WN_HIP(hipGetLastError());
  return 0;
}


int dwconv_ln_silu(const DwConvArgs& a, hipStream_t s) {
  if (tune().dwconv_tiled == 1 && (a.D == 256 || a.D == 512)) {
    dim3 gt(cdiv(cdiv(a.M, 4), 4)), tt(256);
    if (a.D == 256) hipLaunchKernelGGL(dwconv_tiled_kernel<4>, gt, tt, 0, s, a);
    else hipLaunchKernelGGL(dwconv_tiled_kernel<8>, gt, tt, 0, s, a);
    WN_HIP(hipGetLastError());
    return 0;
  }
  dim3 g(cdiv(a.M, 4)), t(256);
#define WN_DW(E)                                              \
  case E * 64:                                                \
    hipLaunchKernelGGL(dwconv_kernel<E>, g, t, 0, s, a);      \
    break;
  switch (a.D) {
    WN_DW(1) WN_DW(2) WN_DW(4) WN_DW(8) WN_DW(12) WN_DW(16) WN_DW(20)
    default:
      set_error("dwconv: unsupported width " + std::to_string(a.D));
      return -1;
  }
#undef WN_DW
  WN_HIP(hipGetLastError());
  return 0;
}


int relpos_fold(float* K, int ldk, const float* P, int ldp, const float* bias_u,
                const float* bias_v, const int* row_utt, const int* off, const int* p_off,
                float* kbias, int n_heads, int M, int D, hipStream_t s) {
  WN_CHECK(D == n_heads * 64 && (D == 256 || D == 512) && ldk % 4 == 0 && ldp % 4 == 0,
           "relpos_fold: shape");
  dim3 g(cdiv(M, 4)), t(256);
  if (D == 256)
    hipLaunchKernelGGL(relpos_fold_kernel<4>, g, t, 0, s, K, ldk, P, ldp, bias_u, bias_v,
                       row_utt, off, p_off, kbias, n_heads, M);
  else
    hipLaunchKernelGGL(relpos_fold_kernel<8>, g, t, 0, s, K, ldk, P, ldp, bias_u, bias_v,
                       row_utt, off, p_off, kbias, n_heads, M);
  WN_HIP(hipGetLastError());
  return 0;
}


int attention(const AttnArgs& a, hipStream_t s) {
  WN_CHECK(a.n_seq > 0 && a.n_heads > 0 && a.max_q_len > 0, "attention: empty");
  WN_CHECK(a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0,
           "attention: strides must be multiples of 4 floats");
  WN_CHECK(a.mask_mode != 2 || a.chunk_size > 0, "attention: chunk size");
  if (t_gemm_prec == PREC_BF16 && tune().attn_bf16 != 0) return attention_bf16(a, s);
  if (t_gemm_prec == PREC_F32 && tune().attn_x6 != 0 && (a.mask_mode == 0 || tune().attn_x6 == 2) &&
      attention_x6_supported(a))
    return attention_x6(a, s);
  constexpr int NW = 2;
  dim3 g(cdiv(a.max_q_len, NW * 32), a.n_heads, a.n_seq), t(NW * 64);
  AttnArgs ax = a;
  if (a.blk_tab && a.n_blk > 0) {
    g = dim3(a.n_blk);       // (a list of 64-query blocks: NW = 2)
  } else if (tune().attn_xcd != 0) {
    ax.xcd_nqb = g.x;
    g = dim3(g.x * g.y * g.z);
  }
  // key split for the encoder's self attention over long sequences: twice the
  // waves for the same tiles (tune().attn_split: 0 auto, 1 off, 2 on)
  const bool split = tune().attn_split == 2 ||
                     (tune().attn_split == 0 && (a.P != nullptr || a.kbias != nullptr) &&
                      a.max_q_len >= 128);
  const bool fold = a.P != nullptr && a.fold && a.bias_u && a.bias_v;
  if (split) {
    dim3 t2(NW * 2 * 64);
    if (fold)
      hipLaunchKernelGGL((attention_kernel<NW, false, 2, true>), g, t2, 0, s, ax);
    else if (a.P)
      hipLaunchKernelGGL((attention_kernel<NW, true, 2>), g, t2, 0, s, ax);
    else
      hipLaunchKernelGGL((attention_kernel<NW, false, 2>), g, t2, 0, s, ax);
  } else if (fold)
    hipLaunchKernelGGL((attention_kernel<NW, false, 1, true>), g, t, 0, s, ax);
  else if (a.P)
    hipLaunchKernelGGL((attention_kernel<NW, true, 1>), g, t, 0, s, ax);
  else
    hipLaunchKernelGGL((attention_kernel<NW, false, 1>), g, t, 0, s, ax);
  WN_HIP(hipGetLastError());
  return 0;
}

int copy_rows(const float* src, int lds, const int* src_rows, float* dst,
              int ldd, const int* dst_rows, int n_rows, int D, hipStream_t s) {
  if (n_rows <= 0) return 0;
  WN_CHECK(D % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0, "copy_rows: % 4");
  hipLaunchKernelGGL(copy_rows_kernel, dim3(n_rows), dim3(64), 0, s, src, lds,
                     src_rows, dst, ldd, dst_rows, n_rows, D / 4);
  WN_HIP(hipGetLastError());
  return 0;
}

// ---- streaming (forward_chunk) cache plumbing -------------------------------
namespace {
// One float4 per thread: frame j of [cache | chunk], head h, 4 of the 128
// (K | V) floats.  Writes the contiguous [Tk][2d] K|V rows the attention
// kernel reads and, for frames >= next_start, the new cache slice
// (heads, new_t1, 128) of this layer (attention.py:207-215, encoder.py:271-279).
__global__ void chunk_kv_kernel(const ChunkSess* __restrict__ sess, int layer,
                                const float* __restrict__ qkv, int R, int H,
                                float* __restrict__ kv) {
  const ChunkSess ss = sess[blockIdx.y];
  const int t1 = ss.t1, d = H * 64, Tk = t1 + R;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Tk * H * 32) return;
  const int e = i & 31, h = (i >> 5) % H, j = i / (32 * H);
  const bool is_v = e >= 16;
  const int c = h * 64 + (e & 15) * 4;
  f32x4 v;
  if (j < t1)
    v = *reinterpret_cast<const f32x4*>(ss.att_cache + ((int64_t)layer * H * t1 +
                                                         (int64_t)h * t1 + j) * 128 + e * 4);
  else
    v = *reinterpret_cast<const f32x4*>(qkv + ((int64_t)blockIdx.y * R + (j - t1)) * 3 * d +
                                        (is_v ? 2 * d : d) + c);
  *reinterpret_cast<f32x4*>(kv + ((int64_t)ss.kv_off + j) * 2 * d + (is_v ? d : 0) + c) = v;
  if (j >= ss.next_start) {
    const int nt = ss.nt;
    *reinterpret_cast<f32x4*>(ss.new_att + ((int64_t)layer * H * nt + (int64_t)h * nt +
                                            (j - ss.next_start)) * 128 + e * 4) = v;
  }
}

// Causal convolution input with its left context (convolution.py:121-130):
// rows [0, lorder) come from the cache ((d, lorder) channel-major; zeros for
// the first chunk), rows lorder.. are the chunk; the last lorder rows are the
// new cache.  Session b owns rows b * (lorder + R) .. of xext.
__global__ void chunk_conv_in_kernel(const ChunkSess* __restrict__ sess, int layer,
                                     const float* __restrict__ x, int R, int d, int lorder,
                                     float* __restrict__ xext) {
  const ChunkSess ss = sess[blockIdx.y];
  const int LR = lorder + R;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= LR * d) return;
  const int row = i / d, c = i - row * d;
  const float* cache = ss.cnn_cache ? ss.cnn_cache + (int64_t)layer * d * lorder : nullptr;
  const float v = row < lorder ? (cache ? cache[(int64_t)c * lorder + row] : 0.f)
                               : x[((int64_t)blockIdx.y * R + (row - lorder)) * d + c];
  xext[(int64_t)blockIdx.y * LR * d + i] = v;
  if (row >= R)
    ss.new_cnn[(int64_t)layer * d * lorder + (int64_t)c * lorder + (row - R)] = v;
}
}  // namespace

int chunk_kv_assemble(const ChunkSess* sess, int n_sess, int layer, int max_tk,
                      const float* qkv, int R, int H, float* kv, hipStream_t s) {
  WN_CHECK(R > 0 && H > 0 && n_sess > 0 && max_tk >= R, "chunk kv: bad argument");
  const int n = max_tk * H * 32;
  hipLaunchKernelGGL(chunk_kv_kernel, dim3(cdiv(n, 256), n_sess), dim3(256), 0, s, sess, layer,
                     qkv, R, H, kv);
  WN_HIP(hipGetLastError());
  return 0;
}

int chunk_conv_input(const ChunkSess* sess, int n_sess, int layer, const float* x, int R,
                     int d, int lorder, float* xext, hipStream_t s) {
  WN_CHECK(R > 0 && lorder > 0 && n_sess > 0, "chunk conv: bad argument");
  const int n = (lorder + R) * d;
  hipLaunchKernelGGL(chunk_conv_in_kernel, dim3(cdiv(n, 256), n_sess), dim3(256), 0, s, sess,
                     layer, x, R, d, lorder, xext);
  WN_HIP(hipGetLastError());
  return 0;
}

int fill_zero(void* p, size_t bytes, hipStream_t s) {
  WN_HIP(hipMemsetAsync(p, 0, bytes, s));
  return 0;
}

}  // namespace wn
