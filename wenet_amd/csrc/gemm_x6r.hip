// Row-block fp32 GEMM on the bf16 matrix cores (six plane products, gemm_x6.hip) for the
// encoder's SMALL projections at d_model = 256 -- K = 256, N = 256 .. 768, M = a few thousand
// rows: the attention output projection and pointwise_conv2 fused with their residual add and
// the LayerNorm that follows (attention.py:176, convolution.py:148; encoder_layer.py:238-240,
// 251-253), and the QKV projection (attention.py:109-131).
//
// Why (round 3): these GEMMs are 1-3 GFLOP each -- 7-20 us of v_mfma_f32 work that took
// 23-37 us per launch (MFMA busy 0.23-0.48, profiles/r03i): tile GEMMs at this size are all
// prologue and epilogue, and the six-product tile GEMM needs a separate pass that writes the
// plane image of A.  Here the structure of ffn_x6f.hip is reused instead:
//   * a block = 4 waves = ONE wave per SIMD owns 32 rows of A; the block loads those rows once
//     as fp32 (whole rows per instruction) and turns them through LDS into the fragment layout
//     (lane = row, its k half); wave w splits k blocks 4 w .. 4 w + 3 into the three bf16 planes
//     and parks the fragments in LDS (48 KB), from where every wave reads the "B" operand of a
//     k block one step ahead of its MFMAs -- no plane image of A in HBM (round 4: the planes
//     were 192 registers of EVERY wave before, split four times over);
//   * the waves split N: wave w computes columns [w N / 4, (w + 1) N / 4) = NT tiles of 32, so
//     each W fragment is read by exactly one wave -- straight from the weight plane image in
//     L2 into registers (16 B per lane, one 1-KB record per instruction, PF k blocks ahead; the
//     six-tile kernels reload a record right behind its last MFMA instead, RELOAD below): no
//     DMA ring, no barrier in the main loop; a CU pulls such a stream at ~52 of the ~55 B/clk
//     it can (tools/probes/stream_probe.hip); 248 blocks re-read the same 0.4-1.2 MB
//     image, which stays in every XCD's L2;
//   * W fragment = the instruction's "A" operand, so a lane ends up with ONE row and the columns
//     8 g + 4 (lane / 32) + q of every tile: a row's LayerNorm statistics are 32 NT values per
//     lane, one exchange with lane ^ 32 and one LDS round between the four waves (two passes:
//     mean, then the centred squares, like layernorm_kernel);
//   * no global access uses lane = row (32 bytes of 32 different lines per instruction: the A
//     rows alone cost ~10 k cycles that way): residual, x_out, y and C tiles pass through a
//     wave-private LDS patch and cross the memory pipe as contiguous row segments (round 3:
//     out-projection + LayerNorm 17.8 -> 14.4 us, QKV 31.6 -> 30.4 us stand-alone).
// Grid = ceil(M / 32) blocks (248 at M = 7932: the 256 CUs once).
#include "common.h"
#include "kernels.h"
#include "rowregs.h"
#include "attn_x6_img.h"
#include "x6.h"

namespace wn {

namespace {

// wave-private LDS patches are written and read back by the SAME wave without a barrier (the
// LDS executes a wave's accesses in order); where the two sides use different vector types the
// compiler must not move one across the other either
#define WAVE_LDS_ORDER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

constexpr int RK = 256;              // K of this kernel
constexpr int RKB = RK / 16;         // 16 k blocks

// EPI 0: C = acc + bias; EPI 1: x_out = resid + alpha (acc + bias), y = LayerNorm(x_out);
// EPI 2: C = GLU(acc + bias) (N / 2 columns); EPI 3: EPI 1, then C = GLU(y W3b^T + bias2) from
// the rows in LDS (y itself is stored only if p.y is set)
// PRO 1: the block forms its A rows itself from the slice partials of the fused feed-forward module
// in front of it (X6RArgs::pro_*): the sum, the residual add and the LayerNorm of ffn_reduce_ln's
// mode 0 -- same operations in the same order, a wave per row, 4 consecutive columns per lane --
// on the rows the prologue holds as whole rows anyway (round 3: one launch and one round trip of
// LN(x) through HBM less per layer).
// shader-clock stamps of the middle block's wave 0 of the last launch, per kernel form: entry, rows in LDS + first W loads +
// barrier, planes in LDS, main loop done, stores drained (wn_tune_set("x6_probe", 8) makes
// wn_profile_gemm_clocks return them; tools/x6r_clocks.py)
__device__ unsigned long long g_x6r_clk[8][8];   // [EPI + 4 (PRO != 0)][stamp]
template <int NT, int EPI, int PF, int PRO = 0>
__global__ __launch_bounds__(256, 1) void x6r_kernel(X6RArgs p) {
  const bool stamp = blockIdx.x == gridDim.x / 2 && threadIdx.x == 0;   // (a block inside an utterance)
  unsigned long long k0 = __builtin_readcyclecounter(), k1 = 0, k2 = 0, k3 = 0, k4 = 0, kp = 0;
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
  __shared__ float red[2][4][32];
  constexpr int PATCH = 4 * 32 * (NT * 128 + 16) > 32 * 1040 ? 4 * 32 * (NT * 128 + 16) : 32 * 1040;
  __shared__ __attribute__((aligned(16))) char patch[PATCH];
  // the three bf16 planes of the block's 32 rows, fragment layout: record (k block, plane) =
  // lane x 16 B (round 4: were 192 registers of EVERY wave, split four times over)
  __shared__ __attribute__((aligned(16))) char ximg[RKB * 3 * 1024];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < 4);
  const int hi = lane >> 5, li = lane & 31;
  const int m0 = blockIdx.x * 32;
  const int row = m0 + li;
  const int rowc = min(row, p.M - 1);

  // ---- A rows: fp32, 8 consecutive floats per k block and lane ---------------------------------
  // The fragment layout wants lane = row -- as a global load that is 64 different 128-byte lines
  // per instruction, and every wave needs all 32 rows.  The block loads the rows ONCE, whole rows
  // per instruction (wave w: rows 8 w .. 8 w + 7), and turns them through LDS: row stride 1040
  // bytes = 260 dwords, conflict-free for the lane = row reads.
  {
    f32x4 rowv[8];
    if constexpr (PRO == 1) {
      const int c = lane * 4;
      const f32x4 b2 = *reinterpret_cast<const f32x4*>(p.pro_b2 + c);
      // the slice loads of a row block IN FLIGHT together: four slices x eight rows per round;
      // per row the additions of ffn_reduce_ln, in its order.  (The first form walked a
      // run-time slice loop per row -- load, s_waitcnt vmcnt(0), add: 8 S dependent memory round
      // trips in front of the GEMM; bit-identical, removed in round 4.)
      f32x4 acc8[8], xo8[8];
      int r8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        r8[j] = min(m0 + wave * 8 + j, p.M - 1);
        acc8[j] = b2;
        xo8[j] = *reinterpret_cast<const f32x4*>(p.pro_x + (int64_t)r8[j] * RK + c);
      }
      int sl = 0;
      for (; sl + 4 <= p.pro_S; sl += 4) {
        f32x4 v[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j)
            v[u][j] = *reinterpret_cast<const f32x4*>(
                p.pro_P + ((int64_t)(sl + u) * p.M + r8[j]) * RK + c);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc8[j] += v[u][j];
      }
      for (; sl < p.pro_S; ++sl) {
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          v[j] = *reinterpret_cast<const f32x4*>(p.pro_P + ((int64_t)sl * p.M + r8[j]) * RK + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc8[j] += v[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) rowv[j][e] = xo8[j][e] + p.pro_alpha * acc8[j][e];
#pragma unroll
      for (int j = 0; j < 8; ++j)       // (rows past M are copies of row M - 1: never stored)
        if (m0 + wave * 8 + j < p.M)
          *reinterpret_cast<f32x4*>(p.pro_x + (int64_t)(m0 + wave * 8 + j) * RK + c) = rowv[j];
      // LayerNorm of the 8 rows (ffn_reduce_ln_kernel's norm(), the eight butterflies side by side)
      float sm[8], sq[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) t += rowv[j][e];
        sm[j] = t;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int j = 0; j < 8; ++j) sm[j] += __shfl_xor(sm[j], o, 64);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sm[j] *= (1.0f / RK);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dd = rowv[j][e] - sm[j];
          q = __builtin_fmaf(dd, dd, q);
        }
        sq[j] = q;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int j = 0; j < 8; ++j) sq[j] += __shfl_xor(sq[j], o, 64);
      const f32x4 gw = *reinterpret_cast<const f32x4*>(p.ln_w + c);
      const f32x4 gb = *reinterpret_cast<const f32x4*>(p.ln_b + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float rstd = 1.0f / sqrtf(sq[j] * (1.0f / RK) + p.eps);
#pragma unroll
        for (int e = 0; e < 4; ++e) rowv[j][e] = (rowv[j][e] - sm[j]) * rstd * gw[e] + gb[e];
      }
    } else if constexpr (PRO == 2) {
      // DWC (round 4): the A rows are the middle of the convolution module -- depthwise conv over
      // time + LayerNorm / eval-BatchNorm affine + SiLU of the GLU output p.dw.x
      // (convolution.py:119-146) -- formed here instead of by dwconv_tiled_kernel: the same
      // operations per output row in the same order (bias, the taps in ascending order with the
      // pad rule evaluated against the output row's own utterance, norm, SiLU; lane = 4
      // consecutive channels = RowRegs<4>); one launch and one round trip of the 8-MB tensor
      // less per layer.  (Same source, not the same bits: this file is compiled with
      // -fno-slp-vectorize and contracts a multiply-add of the shared LayerNorm differently:
      // 3e-6 on unit-scale outputs, tests/test_gpu_ffn_fused.py.)  Wave w: rows 8 w .. 8 w + 7 of the block; a group of 8
      // taps shares its 15 window rows.
      const DwConvArgs& a = p.dw;
      constexpr int R = 8, TG = 8, NWIN = R + TG - 1;
      const int row0 = m0 + wave * R;
      const int lpad = a.causal ? a.K - 1 : (a.K - 1) / 2;
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
      RowRegs<4> acc[R], cp;
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r].load(a.bias, lane);
      cp.load(a.cpad, lane);
      // Branch-free (round 4, second form): the pad rule picks the x operand of ONE fma per (row,
      // tap) -- the window row, the pad row, or 0 (acc + w * 0 = acc: the tap is skipped) --
      // instead of two wave-uniform branches per (row, tap): 128 taken-or-not branches cost more
      // than the 256 fmas they guarded (15.6 k of the prologue's 33 k cycles, r08g).  Same
      // operations on every contributing tap, in the same order.
      for (int k0 = 0; k0 < a.K; k0 += TG) {
        RowRegs<4> wk[TG], xw[NWIN];
#pragma unroll
        for (int i = 0; i < NWIN; ++i) {
          const int q = min(max(row0 + k0 - lpad + i, 0), a.M - 1);
          xw[i].load(a.x + (int64_t)q * a.ldx, lane);
        }
#pragma unroll
        for (int i = 0; i < TG; ++i) {
          const int kq = min(k0 + i, a.K - 1);      // (a tap past K: loaded, never used)
          wk[i].load(a.wt + (int64_t)kq * 256, lane);
        }
        // interior waves (every tap of every row inside its own utterance: all but the ~3 % of
        // waves that touch an utterance boundary) skip the selects altogether
        bool interior = k0 + TG <= a.K;
#pragma unroll
        for (int r = 0; r < R; ++r)
          interior = interior && on[r] && t_r[r] + k0 - lpad >= 0 &&
                     t_r[r] + k0 + TG - 1 - lpad < len_r[r];
        if (interior) {
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < TG; ++j)
#pragma unroll
              for (int e = 0; e < 4; ++e)
                acc[r].v[e] = fmaf(wk[j].v[e], xw[r + j].v[e], acc[r].v[e]);
        } else {
#pragma unroll
          for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int j = 0; j < TG; ++j) {
              const int k = k0 + j, tt = t_r[r] + k - lpad;
              const bool live = on[r] && k < a.K;
              const bool in = live && tt >= 0 && tt < len_r[r];
              const bool pad =
                  live && !in && ((tt < 0 && a.causal) || (tt >= len_r[r] && tt < a.t_max));
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float x = in ? xw[r + j].v[e] : (pad ? cp.v[e] : 0.f);
                acc[r].v[e] = fmaf(wk[j].v[e], x, acc[r].v[e]);
              }
            }
          }
        }
      }
      // norm + SiLU of the eight rows SIDE BY SIDE: ln_inplace's operations per row (rowregs.h),
      // every butterfly stage issued for all rows before the next one -- eight independent chains
      // instead of eight dependent ones of 12 cross-lane steps each (15.4 k cycles, r08g)
      if (a.norm_mode == 0) {
        float sm[R], sq[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          float t = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) t += acc[r].v[e];
          sm[r] = t;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
          for (int r = 0; r < R; ++r) sm[r] += __shfl_xor(sm[r], o, 64);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          sm[r] *= 1.0f / RK;
          float t = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = acc[r].v[e] - sm[r];
            t += d * d;
          }
          sq[r] = t;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
          for (int r = 0; r < R; ++r) sq[r] += __shfl_xor(sq[r], o, 64);
        RowRegs<4> ww, bb;
        ww.load(a.ln_w, lane);
        bb.load(a.ln_b, lane);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float var = sq[r] * (1.0f / RK);
          const float rstd = 1.0f / sqrtf(var + a.eps);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[r].v[e] = (acc[r].v[e] - sm[r]) * rstd * ww.v[e] + bb.v[e];
        }
      } else {
        RowRegs<4> sc, sh;
        sc.load(a.ln_w, lane);
        sh.load(a.ln_b, lane);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[r].v[e] = fmaf(acc[r].v[e], sc.v[e], sh.v[e]);
      }
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) rowv[r][e] = on[r] ? silu_f(acc[r].v[e]) : 0.f;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = min(m0 + wave * 8 + j, p.M - 1);
        rowv[j] = *reinterpret_cast<const f32x4*>(p.A + (int64_t)r * p.lda + lane * 4);
      }
    }
    asm volatile("" : "+v"(rowv[0]), "+v"(rowv[7]));
    kp = __builtin_readcyclecounter();             // (the A rows are in registers)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      *reinterpret_cast<f32x4*>(patch + (wave * 8 + j) * 1040 + lane * 16) = rowv[j];
  }
  // ---- W fragments: records [k block][tile][plane] of the weight image, this wave's NT tiles ---
  const int Tn = (p.N + 31) >> 5;
  const char* wb = reinterpret_cast<const char*>(p.W3) + ((int64_t)wave * NT * 3) * X3_REC +
                   lane * 16;
  const int64_t kstride = (int64_t)Tn * X3_TILE;
  bf16x8 wf[PF + 1][NT][3];
  auto load_w = [&](int ks) {
    const char* q = wb + ks * kstride;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        wf[ks % (PF + 1)][t][pl] = *reinterpret_cast<const bf16x8*>(q + (t * 3 + pl) * X3_REC);
  };
  // PF == 1 (six column tiles: two buffers of 18 records are all the register file leaves): the
  // RELOAD form -- both buffers are filled up front and every record is reloaded (k block ks + 2)
  // right behind its last MFMA of k block ks (plane 2 after the first product, plane 1 after the
  // fourth, plane 0 after the sixth), so that loads are in flight all the time instead of one
  // burst of 18 per k block waited for one k block later: the weight stream of a block ran at
  // 33 B/clk that way against the ~55 B/clk a CU can pull (tools/probes/stream_probe.hip) --
  // 2150 cycles per k block for 1152 cycles of MFMAs (r07v).
  constexpr bool RELOAD = PF == 1;
  auto load_rec = [&](int ks, int t, int pl) {
    wf[ks % (PF + 1)][t][pl] =
        *reinterpret_cast<const bf16x8*>(wb + ks * kstride + (t * 3 + pl) * X3_REC);
  };
  // (behind the row loads and their LDS stores: loads return in order, a W record issued in
  // front of the rows would be waited for with them)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < (RELOAD ? 2 : PF); ++s) load_w(s);
  __syncthreads();
  k1 = __builtin_readcyclecounter();
  // exact three-way bf16 split (x6.h) of the rows in the fp32 patch: wave w splits k blocks
  // 4 w .. 4 w + 3 and parks the fragments in ximg; every wave reads all of them back, one
  // k block ahead of its MFMAs
  auto split_quarter = [&]() {
#pragma unroll
    for (int j = 0; j < RKB / 4; ++j) {
      const int ks = wave * (RKB / 4) + j;
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(patch + li * 1040 + ks * 64 + hi * 32);
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(patch + li * 1040 + ks * 64 + hi * 32 + 16);
      bf16x8 x3[3];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const Split3 sa = split3(a4[e]), sb = split3(b4[e]);
        x3[0][e] = sa.h0; x3[1][e] = sa.h1; x3[2][e] = sa.h2;
        x3[0][4 + e] = sb.h0; x3[1][4 + e] = sb.h1; x3[2][4 + e] = sb.h2;
      }
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        *reinterpret_cast<bf16x8*>(ximg + (ks * 3 + pl) * 1024 + lane * 16) = x3[pl];
    }
  };
  bf16x8 xf[2][3];
  auto load_x = [&](int ks) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      xf[ks & 1][pl] = *reinterpret_cast<const bf16x8*>(ximg + (ks * 3 + pl) * 1024 + lane * 16);
  };
  split_quarter();
  __syncthreads();
  load_x(0);

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // the epilogue's per-column vectors (bias, LayerNorm weight / bias), one batch of loads in
  // flight under the GEMM: as `if (bias) v += bias[c]` inside the epilogue's loops every one of
  // the 24 loads of a six-tile wave was waited for on its own (11 k of the QKV kernel's 54 k
  // cycles, r07v)
  const int ecol = wave * NT * 32 + 4 * hi;
  f32x4 ebias[NT][4], elw[(EPI == 1 || EPI == 3) ? NT : 1][4], elb[(EPI == 1 || EPI == 3) ? NT : 1][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = min(ecol + t * 32 + 8 * g, p.N - 4);
      ebias[t][g] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.bias) ebias[t][g] = *reinterpret_cast<const f32x4*>(p.bias + c);   // (uniform)
      if constexpr (EPI == 1 || EPI == 3) {
        elw[t][g] = *reinterpret_cast<const f32x4*>(p.ln_w + c);
        elb[t][g] = *reinterpret_cast<const f32x4*>(p.ln_b + c);
      }
    }
  k2 = __builtin_readcyclecounter();

  // plane products, the small ones first: (W plane, activation plane); the tiles alternate so
  // that no MFMA waits for its predecessor's accumulator
  constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
  for (int ks = 0; ks < RKB; ++ks) {
    if (!RELOAD && ks + PF < RKB) load_w(ks + PF);
    if (ks + 1 < RKB) load_x(ks + 1);
#pragma unroll
    for (int q = 0; q < 6; ++q) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks % (PF + 1)][t][PW[q]],
                                                         xf[ks & 1][PX[q]], acc[t], 0, 0, 0);
      if (RELOAD && ks + 2 < RKB && (q == 0 || q == 3 || q == 5)) {
        // the plane whose last product this was (PW = {2, 0, 1, 1, 0, 0})
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) load_rec(ks + 2, t, q == 0 ? 2 : q == 3 ? 1 : 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue: lane = row, registers = columns col0 + 32 t + 8 g + 4 hi + e ------------------
  // Global accesses with lane = row move 32 bytes of 32 different lines per instruction; every
  // tile of C / x_out / y and of the residual goes through a wave-private LDS patch instead (32
  // rows x NT * 128 bytes, row stride + 16 bytes: conflict-free for the lane = row side) and
  // crosses the memory pipe as row segments of NT * 128 contiguous bytes.
  asm volatile("" : "+v"(acc[0]), "+v"(acc[NT - 1]));
  k3 = __builtin_readcyclecounter();
  const int col0 = wave * NT * 32;
  constexpr int SEG = NT * 128;                 // bytes of a row this wave owns
  constexpr int PST = SEG + 16;                 // patch row stride
  constexpr int LPR = SEG / 16;                 // lanes per row segment (8 NT)
  constexpr int NIT = 32 * LPR / 64;            // instructions per tile on the coalesced side
  __syncthreads();                              // the A patch is dead
  char* wp = patch + wave * (32 * PST);
  auto put = [&](const f32x4 (&v)[NT][4]) {     // lane = row layout -> patch
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(wp + li * PST + (t * 32 + 8 * g + 4 * hi) * 4) = v[t][g];
  };
  auto get = [&](f32x4 (&v)[NT][4]) {           // patch -> lane = row layout
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        v[t][g] = *reinterpret_cast<const f32x4*>(wp + li * PST + (t * 32 + 8 * g + 4 * hi) * 4);
  };
  // coalesced side: piece q = it * 64 + lane of the tile = row q / LPR, 16-byte piece q % LPR
  auto store_rows = [&](float* base, int ld) {  // patch -> global, whole segments
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = it * 64 + lane, r = q / LPR, pc = q - r * LPR;
      const f32x4 v = *reinterpret_cast<const f32x4*>(wp + r * PST + pc * 16);
      if (m0 + r < p.M && col0 + pc * 4 < p.N)
        *reinterpret_cast<f32x4*>(base + (int64_t)(m0 + r) * ld + col0 + pc * 4) = v;
    }
  };
  if constexpr (EPI == 0) {
    f32x4 v[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        v[t][g] = f32x4{acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]} +
                  ebias[t][g];
      }
    put(v);
    store_rows(p.C, p.ldc);
  } else if constexpr (EPI == 4) {
    // ---- QKV of a layer whose attention reads global-row-aligned key tiles (kernels.h, epi 4):
    // wave = head; tiles 0-1 Q_h, 2-3 K_h, 4-5 V_h of this block's 32 rows = key tile blockIdx.x
    static_assert(EPI != 4 || NT == 6, "epi 4: [Q | K | V] x 64 columns of one head per wave");
    const int head = wave;
    const int u_l = p.at_row_utt[rowc];
    const int prow = rowc - p.at_off[u_l] + (p.at_p_off ? p.at_p_off[u_l] : 0);
    f32x4 fu[2][4], fv[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        fu[t][g] = *reinterpret_cast<const f32x4*>(p.at_u + head * 64 + t * 32 + 8 * g + 4 * hi);
        fv[t][g] = *reinterpret_cast<const f32x4*>(p.at_v + head * 64 + t * 32 + 8 * g + 4 * hi);
      }
    // Q_h: fp32 rows through the patch (lane = row -> whole 256-byte row segments)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(wp + li * PST + (t * 32 + 8 * g + 4 * hi) * 4) =
            f32x4{acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]} +
            ebias[t][g];
    WAVE_LDS_ORDER();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int q = it * 64 + lane, r = q >> 4, pc = q & 15;
      const f32x4 v = *reinterpret_cast<const f32x4*>(wp + r * PST + pc * 16);
      if (m0 + r < p.M)
        *reinterpret_cast<f32x4*>(p.C + (int64_t)(m0 + r) * p.ldc + head * 64 + pc * 4) = v;
    }
    WAVE_LDS_ORDER();
    // the position rows of the block's keys, this head's 64 dims: coalesced -> patch -> lane = row
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int q = it * 64 + lane, r = q >> 4, pc = q & 15;
      const int pr = __shfl(prow, r, 64);
      *reinterpret_cast<f32x4*>(wp + r * PST + pc * 16) =
          *reinterpret_cast<const f32x4*>(p.at_P + (int64_t)pr * p.at_ldp + head * 64 + pc * 4);
    }
    WAVE_LDS_ORDER();
    f32x4 pv[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        pv[t][g] = *reinterpret_cast<const f32x4*>(wp + li * PST + (t * 32 + 8 * g + 4 * hi) * 4);
    WAVE_LDS_ORDER();
    // K_h: the per-key scalar u.k + v.p (the pack pass's pieces of 4 dims and its butterfly
    // order: piece c = 8 t + 2 g + hi; xor 8 = t, xor 4 / 2 = g, xor 1 = the partner lane), then
    // K' = k + p split into planes
    float sc[2][4];
    bf16x4 kp[3][2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 k4 = f32x4{acc[2 + t][4 * g], acc[2 + t][4 * g + 1], acc[2 + t][4 * g + 2],
                         acc[2 + t][4 * g + 3]} + ebias[2 + t][g];
        sc[t][g] = x6_key_scalar4(fu[t][g], k4, fv[t][g], pv[t][g]);
        k4 = k4 + pv[t][g];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const Split3 sp = split3(k4[e]);
          kp[0][t][g][e] = sp.h0; kp[1][t][g][e] = sp.h1; kp[2][t][g][e] = sp.h2;
        }
      }
    float dsum = ((sc[0][0] + sc[1][0]) + (sc[0][2] + sc[1][2])) +
                 ((sc[0][1] + sc[1][1]) + (sc[0][3] + sc[1][3]));
    dsum += __shfl_xor(dsum, 32, 64);
    char* dst = reinterpret_cast<char*>(p.at_img) +
                ((int64_t)blockIdx.x * 4 + head) * AX_IMG_TILE;
    if (hi == 0) *reinterpret_cast<float*>(dst + AX_IMG_BIAS + li * 4) = dsum;
    // K' planes: [32 keys][64 dims] bf16 in the patch (row stride 144 B), out as 16-byte pieces
    constexpr int KROW = 144, KPLB = 32 * KROW;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<bf16x4*>(wp + pl * KPLB + li * KROW + (t * 32 + 8 * g + 4 * hi) * 2) =
              kp[pl][t][g];
    WAVE_LDS_ORDER();
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int q = it * 64 + lane, r = q >> 3, pc = q & 7;
        *reinterpret_cast<f32x4*>(dst + pl * AX_IMG_K + q * 16) =
            *reinterpret_cast<const f32x4*>(wp + pl * KPLB + r * KROW + pc * 16);
      }
    WAVE_LDS_ORDER();
    // V_h^T planes: [64 dims][32 key slots], the image's slot order and group swizzle
    {
      __bf16* vt = reinterpret_cast<__bf16*>(wp);
      const int slot = ax_key_slot(li);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v4 = f32x4{acc[4 + t][4 * g], acc[4 + t][4 * g + 1], acc[4 + t][4 * g + 2],
                                 acc[4 + t][4 * g + 3]} + ebias[4 + t][g];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const Split3 sp = split3(v4[e]);
            const int o = ax_vt_off(t * 32 + 8 * g + 4 * hi + e, slot);
            vt[o] = sp.h0; vt[AX_VPL + o] = sp.h1; vt[2 * AX_VPL + o] = sp.h2;
          }
        }
      WAVE_LDS_ORDER();
#pragma unroll
      for (int it = 0; it < 3 * AX_IMG_V / 1024; ++it) {
        const int q = it * 64 + lane;
        *reinterpret_cast<f32x4*>(dst + 3 * AX_IMG_K + q * 16) =
            *reinterpret_cast<const f32x4*>(wp + q * 16);
      }
    }
  } else if constexpr (EPI == 2) {
    // GLU (convolution.py:117-118) over a weight whose rows are permuted per 64 as [32 a | 32
    // gate] (wn_model_create): tile 2 u of the wave = values, tile 2 u + 1 = their gates; the
    // wave's NT / 2 output tiles are columns col0 / 2 .. of C
    static_assert(EPI != 2 || NT % 2 == 0, "GLU: value / gate tile pairs");
    constexpr int HT = NT / 2, HSEG = HT * 128, HLPR = HSEG / 16, HNIT = 32 * HLPR / 64;
#pragma unroll
    for (int u = 0; u < HT; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = col0 + 2 * u * 32 + 8 * g + 4 * hi;
        f32x4 a = f32x4{acc[2 * u][4 * g], acc[2 * u][4 * g + 1], acc[2 * u][4 * g + 2],
                        acc[2 * u][4 * g + 3]};
        f32x4 gt = f32x4{acc[2 * u + 1][4 * g], acc[2 * u + 1][4 * g + 1],
                         acc[2 * u + 1][4 * g + 2], acc[2 * u + 1][4 * g + 3]};
        a += ebias[2 * u][g];
        gt += ebias[2 * u + 1][g];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = a[e] * wn_rcp(1.0f + wn_exp(-gt[e]));
        *reinterpret_cast<f32x4*>(wp + li * PST + (u * 32 + 8 * g + 4 * hi) * 4) = o;
      }
#pragma unroll
    for (int it = 0; it < HNIT; ++it) {
      const int q = it * 64 + lane, r = q / HLPR, pc = q - r * HLPR;
      const f32x4 v = *reinterpret_cast<const f32x4*>(wp + r * PST + pc * 16);
      if (m0 + r < p.M)
        *reinterpret_cast<f32x4*>(p.C + (int64_t)(m0 + r) * p.ldc + col0 / 2 + pc * 4) = v;
    }
  } else {
    f32x4 v[NT][4], rs[NT][4];
    // (loading these rows in the prologue instead, behind the A rows, was slower: 14.4 -> 15.8 us)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = it * 64 + lane, r = q / LPR, pc = q - r * LPR;
      const int rc = min(m0 + r, p.M - 1);
      *reinterpret_cast<f32x4*>(wp + r * PST + pc * 16) =
          *reinterpret_cast<const f32x4*>(p.resid + (int64_t)rc * p.ldr + col0 + pc * 4);
    }
    get(rs);
    float s1 = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 a = f32x4{acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2],
                              acc[t][4 * g + 3]} + ebias[t][g];
        v[t][g] = rs[t][g] + p.alpha * a;
        s1 += (v[t][g][0] + v[t][g][1]) + (v[t][g][2] + v[t][g][3]);
      }
    put(v);
    store_rows(p.x_out, p.ldx);
    // mean over the row's N columns: lane pair, then the four waves
    s1 += __shfl_xor(s1, 32, 64);
    if (hi == 0) red[0][wave][li] = s1;
    __syncthreads();
    const float mean =
        ((red[0][0][li] + red[0][1][li]) + (red[0][2][li] + red[0][3][li])) / (float)p.N;
    float s2 = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[t][g][e] - mean;
          s2 += d * d;
        }
    s2 += __shfl_xor(s2, 32, 64);
    if (hi == 0) red[1][wave][li] = s2;
    __syncthreads();
    const float var =
        ((red[1][0][li] + red[1][1][li]) + (red[1][2][li] + red[1][3][li])) / (float)p.N;
    const float rstd = 1.0f / sqrtf(var + p.eps);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 w = elw[t][g], b = elb[t][g];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[t][g][e] = (v[t][g][e] - mean) * rstd * w[e] + b[e];
      }
    if (p.y != nullptr) {
      put(v);
      store_rows(p.y, p.ldy);
    }
    if constexpr (EPI == 1) {
      if (p.y3 != nullptr) {
        // y as its X3 plane image (x6.h x3_store_tile): the consumer -- the fused feed-forward
        // module -- sees the same fragments it would have made of the fp32 rows
        const int tiles_m = (p.M + 31) >> 5;
#pragma unroll
        for (int t = 0; t < NT; ++t)
          x3_store_tile(v[t], p.y3, (col0 + t * 32) >> 4, tiles_m, blockIdx.x, hi, li);
      }
    }
    if constexpr (EPI == 3) {
      // ---- chained: C = GLU(y W2^T + bias2), N2 = 512 (encoder_layer.py:240-251 /
      // convolution.py:115-118): the LayerNorm rows never leave the CU -- every wave puts its 64
      // columns of y into the shared A patch, reads the full rows back in the fragment layout,
      // splits them, and the second GEMM runs like the first (wave w: tiles 4 w .. 4 w + 3 of
      // the [32 values | 32 gates] image)
      __syncthreads();                            // the wave patches are dead
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(patch + li * 1040 + (col0 + t * 32 + 8 * g + 4 * hi) * 4) =
              v[t][g];
      constexpr int NT2 = 4, PF2 = 2;
      const char* wb2 = reinterpret_cast<const char*>(p.W3b) +
                        ((int64_t)wave * NT2 * 3) * X3_REC + lane * 16;
      const int64_t kstride2 = (int64_t)16 * X3_TILE;       // 512 / 32 tiles per k block
      bf16x8 wf2[PF2 + 1][NT2][3];
      auto load_w2 = [&](int ks) {
        const char* q = wb2 + ks * kstride2;
#pragma unroll
        for (int t = 0; t < NT2; ++t)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            wf2[ks % (PF2 + 1)][t][pl] =
                *reinterpret_cast<const bf16x8*>(q + (t * 3 + pl) * X3_REC);
      };
#pragma unroll
      for (int s2i = 0; s2i < PF2; ++s2i) load_w2(s2i);
      __syncthreads();
      split_quarter();                            // (ximg: the first GEMM is done with it)
      __syncthreads();
      load_x(0);
      f32x16 acc2[NT2];
#pragma unroll
      for (int t = 0; t < NT2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;
      f32x4 ebias2[NT2][4];                        // (one batch, under the GEMM: see ebias)
#pragma unroll
      for (int t = 0; t < NT2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          ebias2[t][g] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (p.bias2)
            ebias2[t][g] = *reinterpret_cast<const f32x4*>(p.bias2 + wave * NT2 * 32 + t * 32 +
                                                           8 * g + 4 * hi);
        }
#pragma unroll
      for (int ks = 0; ks < RKB; ++ks) {
        if (ks + PF2 < RKB) load_w2(ks + PF2);
        if (ks + 1 < RKB) load_x(ks + 1);
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int t = 0; t < NT2; ++t)
            acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf2[ks % (PF2 + 1)][t][PW[q]],
                                                              xf[ks & 1][PX[q]], acc2[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();                            // the A patch is dead
      const int c2 = wave * NT2 * 32;             // first column of the wave's tiles in the image
      char* wp2 = patch + wave * (32 * 272);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = c2 + 2 * u * 32 + 8 * g + 4 * hi;
          f32x4 a = f32x4{acc2[2 * u][4 * g], acc2[2 * u][4 * g + 1], acc2[2 * u][4 * g + 2],
                          acc2[2 * u][4 * g + 3]};
          f32x4 gt = f32x4{acc2[2 * u + 1][4 * g], acc2[2 * u + 1][4 * g + 1],
                           acc2[2 * u + 1][4 * g + 2], acc2[2 * u + 1][4 * g + 3]};
          a += ebias2[2 * u][g];
          gt += ebias2[2 * u + 1][g];
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = a[e] * wn_rcp(1.0f + wn_exp(-gt[e]));
          *reinterpret_cast<f32x4*>(wp2 + li * 272 + (u * 32 + 8 * g + 4 * hi) * 4) = o;
        }
#pragma unroll
      for (int it = 0; it < 8; ++it) {              // 32 rows x 16 pieces of 16 bytes
        const int q = it * 64 + lane, r = q >> 4, pc = q & 15;
        const f32x4 o = *reinterpret_cast<const f32x4*>(wp2 + r * 272 + pc * 16);
        if (m0 + r < p.M)
          *reinterpret_cast<f32x4*>(p.C + (int64_t)(m0 + r) * p.ldc + c2 / 2 + pc * 4) = o;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  k4 = __builtin_readcyclecounter();
  if (stamp) {
    unsigned long long* o = g_x6r_clk[(EPI == 4 ? 0 : EPI) + (PRO != 0 ? 4 : 0)];
    o[0] = k0; o[1] = k1; o[2] = k2; o[3] = k3; o[4] = k4; o[5] = kp;
    o[6] = rt0; o[7] = __builtin_amdgcn_s_memrealtime();
  }
}

template <int NT, int EPI, int PF, int PRO = 0>
int launch_x6r(const X6RArgs& a, hipStream_t s) {
  hipLaunchKernelGGL((x6r_kernel<NT, EPI, PF, PRO>), dim3(cdiv(a.M, 32)), dim3(256), 0, s, a);
  WN_HIP(hipGetLastError());
  return 0;
}

}  // namespace

int gemm_x6r_clocks(unsigned long long* out) {
  WN_HIP(hipDeviceSynchronize());
  WN_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_x6r_clk), sizeof(g_x6r_clk)));
  return 0;
}

bool gemm_x6r_supported(int M, int N, int K, int epi) {
  if (K == 512) return gemm_x6r512_supported(M, N, epi);
  if (K != RK || M <= 0) return false;
  if (epi == 1 || epi == 3) return N == 256;
  if (epi == 2) return N == 512;
  if (epi == 4) return N == 768;
  return N == 256 || N == 512 || N == 768;
}

int gemm_x6r(const X6RArgs& a, hipStream_t s) {
  if (a.K == 512) return gemm_x6r512(a, s);
  WN_CHECK(a.K == RK, "gemm_x6r: K must be 256 or 512");
  if (a.pro_P) {
    WN_CHECK((a.epi == 0 || a.epi == 4) && a.N == 768 && a.W3 && a.M > 0 && a.pro_S >= 1 &&
                 a.pro_b2 && a.pro_x && a.ln_w && a.ln_b && a.C && a.ldc % 4 == 0,
             "gemm_x6r: prologue fold arguments");
    if (a.epi == 4) {
      WN_CHECK(a.at_img && a.at_P && a.at_ldp % 4 == 0 && a.at_u && a.at_v && a.at_row_utt &&
                   a.at_off && a.bias, "gemm_x6r: key-tile image arguments (epi 4)");
      return launch_x6r<6, 4, 1, 1>(a, s);
    }
    return launch_x6r<6, 0, 1, 1>(a, s);
  }
  WN_CHECK((a.A || a.dw_on) && a.W3 && a.lda % 4 == 0 && a.epi != 4 &&
               gemm_x6r_supported(a.M, a.N, RK, a.epi), "gemm_x6r: shape");
  if (a.dw_on) {
    WN_CHECK(a.epi == 1 && a.N == 256 && a.W3 && a.M > 0 && a.dw.D == 256 && a.dw.M == a.M &&
                 a.dw.x && a.dw.ldx % 4 == 0 && a.dw.wt && a.dw.bias && a.dw.cpad && a.dw.ln_w &&
                 a.dw.ln_b && a.dw.row_utt && a.dw.off && a.dw.len && a.dw.K >= 1 &&
                 a.resid && a.x_out && a.ln_w && a.ln_b && (a.y || a.y3) && a.ldr % 4 == 0 &&
                 a.ldx % 4 == 0 && a.ldy % 4 == 0, "gemm_x6r: depthwise-conv prologue arguments");
    return launch_x6r<2, 1, 3, 2>(a, s);
  }
  if (a.epi == 1) {
    WN_CHECK(a.resid && a.x_out && a.ln_w && a.ln_b && (a.y || a.y3) && a.ldr % 4 == 0 && a.ldx % 4 == 0 &&
                 a.ldy % 4 == 0, "gemm_x6r: row-LN epilogue arguments");
    return launch_x6r<2, 1, 3>(a, s);
  }
  if (a.epi == 3) {
    WN_CHECK(a.resid && a.x_out && a.ln_w && a.ln_b && a.ldr % 4 == 0 && a.ldx % 4 == 0 &&
                 (a.y == nullptr || a.ldy % 4 == 0) && a.W3b && a.C && a.ldc % 4 == 0,
             "gemm_x6r: chained row-LN + GLU arguments");
    return launch_x6r<2, 3, 3>(a, s);
  }
  WN_CHECK(a.C && a.ldc % 4 == 0, "gemm_x6r: no output");
  if (a.epi == 2) return launch_x6r<4, 2, 2>(a, s);
  switch (a.N) {
    case 256: return launch_x6r<2, 0, 3>(a, s);
    case 512: return launch_x6r<4, 0, 2>(a, s);
    default: return launch_x6r<6, 0, 1>(a, s);
  }
}

}  // namespace wn
