// Fused feed-forward module as fp32 on the bf16 matrix cores (six plane products, gemm_x6.hip)
// with the hidden tensor ON CHIP: P[s] = act(X W1_s^T + b1_s) W2_s^T for the hidden slice s of
// the block, X = LayerNorm(x) -- PositionwiseFeedForward.forward,
// wenet/models/transformer/positionwise_feed_forward.py:50-58 (w_2(act(w_1 x))); the caller's
// next kernel (ffn_reduce_ln) adds the S slice partials, b2 and the residual and applies the
// following LayerNorm (encoder_layer.py:220-228,253-263).
//
// Why (round 3): the two six-product FFN GEMMs of gemm_x6.hip ran at 0.36-0.39 of the bf16
// matrix pipe because the (M, 2048) hidden tensor left the chip as a 6-byte-per-element plane
// image (97.5 MB written by w_1, 104 MB read back by w_2: ~200 MB of HBM round trip per module,
// 4.8 GB per decode step; profiles/r02kc).  Here it never exists in memory -- not even in LDS.
//
// Structure: d_model = 256.  A block = 4 waves = ONE wave per SIMD with the whole 512-entry
// register file each; wave w owns 32 rows of X for the whole launch:
//   * its X rows as MFMA operand fragments in registers for the whole kernel: 16 k blocks x
//     3 planes x 4 registers = 192 registers (fp32 rows of LN(x) read once, split in registers);
//   * Y[32 rows][256] fp32 = 8 accumulator tiles = 128 registers, resident for the whole block;
//   * per chunk of 64 hidden units: phase A  H^T[64][32 rows] = W1[chunk] X^T  (K = 256, the W1
//     fragment is the instruction's "A" operand, so a lane ends up with ONE row m of X and the
//     hidden units n = 8 g + 4 (lane / 32) + q of each 32-unit tile), bias + activation + exact
//     three-way bf16 split in registers, and phase B  Y^T[256][32 rows] += W2[:, chunk] H^T
//     (K = 64) in which the lane's own 8 values per 16-unit k block ARE its "B" operand
//     fragment -- provided k slot j of lane half h means hidden unit 8 (j / 4) + 4 h + j % 4 of
//     the block, which is a property of the W2 image only: `x6_split_perm` builds it once per
//     model.  No exchange between lanes, no LDS round trip, no barrier between the phases.
// The four waves share nothing but the weight stream: W1 / W2 plane records (1 KB = 32 rows x
// 16 k of one plane, the lane order of an MFMA operand) come global -> LDS by DMA
// (buffer_load_dwordx4 ... lds) in stages of 24 records, ring of RING stages, ONE barrier per
// stage placed between the two halves of the stage's last MFMA group; every wave reads every
// record (ds_read_b128, linear, conflict-free).  Per chunk and wave: 384 MFMAs (12288 matrix
// cycles) against 8 x 24 KB of DMA = 16 B / clk / CU, 50 % of the LDS read rate, and the
// bias / SiLU / split VALU work (one 2-element piece per MFMA group) slotted between MFMAs of
// stages that do not depend on it: the split of hidden tile 0 runs under the phase-A MFMAs of
// tile 1, the split of tile 1 under the first half of phase B.
// Grid: tiles_m x S blocks (S hidden slices) filling the 256 CUs once; block -> (slice, row
// tile) so that the blocks one XCD runs share ONE slice of W1 / W2 (1.6 MB of planes at S = 4)
// in its L2.
//
// Arithmetic: identical products to gemm_x6.hip (six of the nine plane products, small ones
// first); what differs from the two-GEMM path is only the fp32 summation order (two
// accumulators per hidden tile over even / odd k blocks; the hidden sum per 16-unit block).
#include <type_traits>

#include "common.h"
#include "gemm_epilogue.h"
#include "kernels.h"
#include "x6.h"

namespace wn {

namespace {

// VAR & 8192 (measurement): shader-clock stamps of one block's waves at the sub-stage boundaries
// of its last steady-state chunk (tools/bench_x6.py --clocks)
__device__ unsigned long long g_x6f_clk[4][24];

constexpr int REC = X3_REC;
constexpr int XSTAGE = 24 * REC;     // 24 records per stage
constexpr int XD = 256;              // d_model of this kernel
constexpr int XKB = XD / 16;         // 16 k blocks of X

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Frag6 { bf16x8 p[2][3]; };    // two operand tiles x three planes

// fp32 [R][ld] (K columns) -> X3 image whose k slots inside a 16-k block are permuted for a
// "B"-operand that a lane assembles from its own accumulator registers (see the header): slot j
// of half h holds column 16 kb + 8 (j / 4) + 4 h + j % 4.
__global__ __launch_bounds__(256) void x6_split_perm_kernel(const float* __restrict__ src, int R,
                                                            int K, int ld,
                                                            char* __restrict__ dst) {
  const int tiles = (R + 31) >> 5, nkb = K >> 4;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int row_l = (int)(idx & 31), h = (int)((idx >> 5) & 1);
  const int64_t rest = idx >> 6;
  const int kb = (int)(rest % nkb), tile = (int)(rest / nkb);
  if (tile >= tiles) return;
  const int row = tile * 32 + row_l;
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
  if (row < R) {
    const float* s = src + (int64_t)row * ld + kb * 16 + h * 4;
    a = *reinterpret_cast<const f32x4*>(s);
    b = *reinterpret_cast<const f32x4*>(s + 8);
  }
  bf16x8 p0, p1, p2;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const Split3 sa = split3(a[e]), sb = split3(b[e]);
    p0[e] = sa.h0; p1[e] = sa.h1; p2[e] = sa.h2;
    p0[4 + e] = sb.h0; p1[4 + e] = sb.h1; p2[4 + e] = sb.h2;
  }
  char* o = dst + ((int64_t)kb * tiles + tile) * X3_TILE + h * 512 + row_l * 16;
  *reinterpret_cast<bf16x8*>(o) = p0;
  *reinterpret_cast<bf16x8*>(o + REC) = p1;
  *reinterpret_cast<bf16x8*>(o + 2 * REC) = p2;
}

// counted wait: `stages` younger stages (6 DMA pieces of this wave each) may stay in flight
__device__ __forceinline__ void wait_vm_stages(int stages) {
  switch (stages) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
  }
}

// VAR: measurement variants (wn_tune_set("ffn_x6f_var"), tools/bench_x6.py; wrong results by
// design except 0 and 16): 1 no MFMAs (reads kept), 2 no DMA inside the loop, 4 no bias /
// activation / split pieces, 8 no waits / barrier inside the loop, 64 no fragment reads; 128 /
// 256 = partials stored sc0 sc1 / nt; 512 = stages of 48 records (ring of 3).
template <int ACT, int RING, int VAR>
__global__ __launch_bounds__(256, 1) void ffn_x6f_kernel(FfnX6Args p, int tiles_m, int pairmap) {
  extern __shared__ __attribute__((aligned(16))) char smem_g[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  // VAR & 512: stages of 48 records (two "sub-stages" of 24: one barrier per 96 MFMAs of a wave)
  constexpr int HALF = (VAR & 512) ? 2 : 1;
  constexpr int STG = XSTAGE * HALF;             // bytes per stage
  constexpr int SPC = 8 / HALF;                  // stages per chunk
  static_assert(RING >= 3 && RING * STG <= 6 * XSTAGE, "ring of at most 144 KB");
  float* b1s = reinterpret_cast<float*>(smem_g + RING * STG);

  // block -> (row tile, hidden slice): block b runs on XCD b % 8; the blocks of an XCD share
  // one slice (S <= 8) or two (S = 16)
  const int bid = blockIdx.x, xcd = bid & 7, rr = bid >> 3;
  int slice, tm;
  if (pairmap && p.S >= 2 && p.S <= 8) {
    // two slices per XCD: the two blocks of a row tile that share an XCD (consecutive rr) fetch
    // its X rows from HBM once -- 16 MB instead of 32 at config 2 -- and the XCD's L2 holds
    // two W slices (3 MB)
    const int nsp = p.S >> 1, pair = xcd % nsp, tg = xcd / nsp;
    slice = 2 * pair + (rr & 1);
    tm = (rr >> 1) * (8 / nsp) + tg;
  } else if (p.S <= 8) {
    slice = xcd & (p.S - 1);
    tm = rr * (8 / p.S) + xcd / p.S;
  } else {
    const int sub = p.S >> 3;
    slice = xcd + 8 * (rr % sub);
    tm = rr / sub;
  }
  if (tm >= tiles_m) return;

  unsigned long long clk[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if constexpr ((VAR & 8192) != 0) {
    clk[9] = __builtin_readcyclecounter();
    clk[12] = __builtin_amdgcn_s_memrealtime();
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < 4);
  const int hi = lane >> 5, li = lane & 31;
  const int NC = p.F / p.S / 64;                 // chunks of 64 hidden units in this block
  const int cg0 = slice * NC;                    // first chunk (global index)
  const int FT = p.F >> 5;                       // 32-row tiles of the W1 image
  const unsigned vlane = (unsigned)lane * 16u;

  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.W13)), 0,
      (int)((int64_t)p.F * XD * 6), 0x00020000);
  const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.W2p)), 0,
      (int)((int64_t)p.F * XD * 6), 0x00020000);

  // ---- stage t = 8 c + sub of the block ---------------------------------------------------
  // sub 0..3: W1 records of hidden tile nt = sub / 2 of chunk c, k blocks (sub & 1) * 8 .. + 8,
  // stored [k block][plane]; sub 4..7: W2 records of hidden k block 4 (cg0 + c) + sub - 4, all 8
  // d tiles, stored [d tile][plane].  Either way MFMA group g of the stage (two operand tiles)
  // reads records 6 g .. 6 g + 5, and wave w issues the DMA of records 6 w .. 6 w + 5.
  auto issue_sub = [&](int c, int sub, int buf) {
    char* dst = smem_g + buf * STG + (sub % HALF) * XSTAGE + wave * 6 * REC;
    const int cg = cg0 + c;
    if (sub < 4) {
      const int nt = sub >> 1;
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2) {
        const int ks = (sub & 1) * 8 + wave * 2 + j2;
        const int so = ((ks * FT + 2 * cg + nt) * 3) * REC;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_ptr)(dst + (j2 * 3 + pl) * REC), 16,
                                                   vlane, so + pl * REC, 0, 0);
      }
    } else {
      const int so = ((cg * 4 + sub - 4) * 24 + wave * 6) * REC;
#pragma unroll
      for (int j = 0; j < 6; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (lds_ptr)(dst + j * REC), 16, vlane,
                                                 so + j * REC, 0, 0);
    }
  };
  // stage st (0..SPC-1) of chunk c = sub-stages st * HALF .. + HALF
  auto issue = [&](int c, int st, int buf) {
#pragma unroll
    for (int h = 0; h < HALF; ++h) issue_sub(c, st * HALF + h, buf);
  };
  // piece j (0..5) of this wave's share of a sub-stage
  auto issue_one = [&](int c, int sub, int buf, int j) {
    char* dst = smem_g + buf * STG + (sub % HALF) * XSTAGE + wave * 6 * REC + j * REC;
    const int cg = cg0 + c;
    if (sub < 4) {
      const int nt = sub >> 1, j2 = j / 3, pl = j - j2 * 3;
      const int ks = (sub & 1) * 8 + wave * 2 + j2;
      const int so = ((ks * FT + 2 * cg + nt) * 3) * REC;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_ptr)dst, 16, vlane, so + pl * REC, 0, 0);
    } else {
      const int so = ((cg * 4 + sub - 4) * 24 + wave * 6) * REC;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (lds_ptr)dst, 16, vlane, so + j * REC, 0, 0);
    }
  };
  auto read_group = [&](int buf, int g) {
    const char* st = smem_g + buf * STG + g * 6 * REC + lane * 16;
    Frag6 f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        f.p[t][pl] = *reinterpret_cast<const bf16x8*>(st + (t * 3 + pl) * REC);
    return f;
  };

  // ---- prologue: X fragments, bias slice, the first RING stages ------------------------------
  // X rows as fp32.  The fragment layout wants lane = row (li of the wave's 32 rows, k half hi: 8
  // consecutive floats per k block) -- but a global load with lane = row touches 64 different
  // 128-byte lines per instruction: 32 such loads per wave took ~10 k cycles of this prologue
  // (r03x stamps).  So the rows are loaded COALESCED -- pass q = k blocks 4 q .. 4 q + 3 = a
  // 256-byte segment of every row, instruction j = rows 4 j .. 4 j + 3, 16 lanes per segment --
  // and turned into the fragment layout through the ring buffer that is still idle (the DMA of
  // stage RING - 1 starts behind the first barrier): a wave-private 32 x (256 + 16)-byte patch,
  // row stride 68 dwords = conflict-free ds_read_b128 for lane = row.
  constexpr bool XT = (VAR & 16384) != 0;
  // VAR & 32768: X arrives as its plane image (FfnX6Args::X3): the wave's 32 rows are row tile
  // 4 tm + wave, every (k block, plane) record of it one 1-KB load -- 48 loads straight into
  // the fragment registers, no turn through LDS, no split (6.6 k of the prologue's 13.8 k
  // cycles, done once by the producer instead of by each of the S slice blocks)
  constexpr bool XI = (VAR & 32768) != 0;
  bf16x8 X[XKB][3];
  const int xrow = min(tm * 128 + wave * 32 + li, p.M - 1);
  f32x4 xa[XKB], xb[XKB];
  f32x4 xr[4][8];
  if constexpr (XI) {
    const int tiles = (p.M + 31) >> 5;
    const int tile = min(tm * 4 + wave, tiles - 1);   // (a tile past the end: rows never stored)
    const char* xi = reinterpret_cast<const char*>(p.X3) + (int64_t)tile * X3_TILE + lane * 16;
    const int64_t kstr = (int64_t)tiles * X3_TILE;
#pragma unroll
    for (int ks = 0; ks < XKB; ++ks)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        X[ks][pl] = *reinterpret_cast<const bf16x8*>(xi + ks * kstr + pl * REC);
  } else if constexpr (XT) {
    const int r4 = lane >> 4, c16 = lane & 15;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = min(tm * 128 + wave * 32 + 4 * j + r4, p.M - 1);
        xr[q][j] = *reinterpret_cast<const f32x4*>(p.X + (int64_t)row * p.ldx + q * 64 + c16 * 4);
      }
  } else {
    const float* xr0 = p.X + (int64_t)xrow * p.ldx + hi * 8;
#pragma unroll
    for (int ks = 0; ks < XKB; ++ks) {
      xa[ks] = *reinterpret_cast<const f32x4*>(xr0 + ks * 16);
      xb[ks] = *reinterpret_cast<const f32x4*>(xr0 + ks * 16 + 4);
    }
  }
#pragma unroll
  for (int t = 0; t < ((VAR & 16384) ? RING - 1 : RING); ++t) issue(0, t, t);     // (RING <= SPC)
  // the bias slice: loaded behind the X rows and the first stages (the counter that orders
  // them retires in issue order -- a wait for these values in front of the DMA issue would hold
  // the DMA back until every X row has arrived), stored to LDS after the split
  // (clamped, unconditional: `i < n ? b1[i] : 0` made each of them a branch with its own wait)
  float b1v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) b1v[i] = p.b1[cg0 * 64 + min(tid + 256 * i, NC * 64 - 1)];
  if constexpr ((VAR & 8192) != 0) { __builtin_amdgcn_sched_barrier(0); clk[15] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
  // exact three-way bf16 split of the rows in registers (x6.h): the operand fragments of
  // phase A for the whole launch; no plane image of LN(x) is ever written
  if constexpr (XT && !XI) {
    char* patch = smem_g + (RING - 1) * STG + wave * (32 * 272);
    const int r4 = lane >> 4, c16 = lane & 15;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        *reinterpret_cast<f32x4*>(patch + (4 * j + r4) * 272 + c16 * 16) = xr[q][j];
      // (wave-private patch: the wave's own LDS operations complete in order)
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        xa[4 * q + k4] = *reinterpret_cast<const f32x4*>(patch + li * 272 + k4 * 64 + hi * 32);
        xb[4 * q + k4] = *reinterpret_cast<const f32x4*>(patch + li * 272 + k4 * 64 + hi * 32 + 16);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the next pass overwrites the patch
    }
  }
  if constexpr (!XI) {
#pragma unroll
    for (int ks = 0; ks < XKB; ++ks) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const Split3 sa = split3(xa[ks][e]), sb = split3(xb[ks][e]);
        X[ks][0][e] = sa.h0; X[ks][1][e] = sa.h1; X[ks][2][e] = sa.h2;
        X[ks][0][4 + e] = sb.h0; X[ks][1][4 + e] = sb.h1; X[ks][2][4 + e] = sb.h2;
      }
    }
  }
  if constexpr ((VAR & 8192) != 0) { __builtin_amdgcn_sched_barrier(0); clk[16] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr ((VAR & 8192) != 0) { __builtin_amdgcn_sched_barrier(0); clk[17] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
  for (int ks = 0; ks < XKB; ++ks)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) asm volatile("" : "+v"(X[ks][pl]));
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (tid + 256 * i < NC * 64) b1s[tid + 256 * i] = b1v[i];
  __syncthreads();
  if constexpr ((VAR & 8192) != 0) { __builtin_amdgcn_sched_barrier(0); clk[18] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }

  f32x16 Y[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) Y[i][r] = 0.f;
  // hidden accumulators: tile nt over the even (E) / odd (O) k blocks of X -- two independent
  // MFMA chains per tile; the bias / activation / split pieces add the two halves
  f32x16 HE[2], HO[2];
  u32x4 Hp[4][3];       // planes of the activated hidden chunk: [16-unit k block][plane]
  if (VAR & 4)
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      Hp[i / 3][i % 3] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
      asm volatile("" : "+v"(Hp[i / 3][i % 3]));
    }

  // plane products, the small ones first: (W plane, activation plane)
  constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};
  // fragment read order of a group: the planes in the order the products need them
  constexpr int RT[6] = {0, 1, 0, 1, 0, 1}, RP[6] = {2, 2, 0, 0, 1, 1};

  // Bias + activation + exact three-way bf16 split of the lane's hidden values, two elements
  // (r, r + 1) per "piece", a piece cut into six slices that ride on the six MFMA pairs of a
  // group: elements r of tile nt = hidden units (r & 3) + 8 (r >> 2) + 4 hi of the tile ->
  // slot r & 7 of k block 2 nt + (r >> 3).
  float pv0 = 0.f, pv1 = 0.f, pt0 = 0.f, pt1 = 0.f;
  unsigned ph0 = 0, ph1 = 0;
  auto pk = [](float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 h = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, h);
  };
  auto lo = [](unsigned u) { return __builtin_bit_cast(float, u << 16); };
  auto hi16 = [](unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); };
  // (every slice starts with an empty volatile asm on the values it continues from: the slices
  // are pure arithmetic and would otherwise all be emitted in front of the group's MFMAs; the
  // asm chains them to the sched_barriers between the MFMAs.)  Twelve slices, one per MFMA of
  // a group, ~3 VALU instructions each: with ONE wave per SIMD an MFMA covers 32 cycles of
  // this wave's own issue and nothing else does.
  auto piece_slice = [&](int nt, int i, int k, const f32x4& bq) {
    const int r = 2 * i;
    switch (k) {
      case 0: {
        float b0 = bq[r & 3];
        asm volatile("" : "+v"(b0));
        pv0 = HE[nt][r] + b0;
        break;
      }
      case 1: {
        float b1 = bq[(r + 1) & 3];
        asm volatile("" : "+v"(b1), "+v"(pv0));
        pv0 += HO[nt][r];
        pv1 = HE[nt][r + 1] + b1;
        break;
      }
      case 2:
        asm volatile("" : "+v"(pv0), "+v"(pv1));
        pv1 += HO[nt][r + 1];
        if (ACT == ACT_SILU) pt0 = pv0 * -1.4426950408889634f;
        break;
      case 3:
        if (ACT == ACT_SILU) {
          asm volatile("" : "+v"(pt0), "+v"(pv1));
          pt1 = pv1 * -1.4426950408889634f;
          pt0 = wn_exp2(pt0);
        }
        break;
      case 4:
        if (ACT == ACT_SILU) {
          asm volatile("" : "+v"(pt0), "+v"(pt1));
          pt1 = wn_exp2(pt1);
          pt0 = 1.0f + pt0;
        }
        break;
      case 5:
        if (ACT == ACT_SILU) {
          asm volatile("" : "+v"(pt0), "+v"(pt1));
          pt1 = 1.0f + pt1;
          pt0 = wn_rcp(pt0);
        }
        break;
      case 6:
        asm volatile("" : "+v"(pv0));
        if (ACT == ACT_SILU) {
          asm volatile("" : "+v"(pt0), "+v"(pt1));
          pt1 = wn_rcp(pt1);
          pv0 *= pt0;
        }
        if (ACT == ACT_RELU) pv0 = fmaxf(pv0, 0.0f);
        break;
      case 7:
        asm volatile("" : "+v"(pv0), "+v"(pv1));
        if (ACT == ACT_SILU) {
          asm volatile("" : "+v"(pt1));
          pv1 *= pt1;
        }
        if (ACT == ACT_RELU) pv1 = fmaxf(pv1, 0.0f);
        ph0 = pk(pv0, pv1);
        break;
      case 8:
        asm volatile("" : "+v"(pv0), "+v"(pv1), "+v"(ph0));
        pv0 -= lo(ph0); pv1 -= hi16(ph0);
        break;
      case 9:
        asm volatile("" : "+v"(pv0), "+v"(pv1));
        ph1 = pk(pv0, pv1);
        break;
      case 10:
        asm volatile("" : "+v"(pv0), "+v"(pv1), "+v"(ph1));
        pv0 -= lo(ph1); pv1 -= hi16(ph1);
        break;
      default: {
        asm volatile("" : "+v"(pv0), "+v"(pv1));
        const int kb = 2 * nt + (r >> 3), d = (r & 7) >> 1;
        Hp[kb][0][d] = ph0;
        Hp[kb][1][d] = ph1;
        Hp[kb][2][d] = pk(pv0, pv1);
        break;
      }
    }
  };
  auto read_frag = [&](int b, int g, int t, int pl) {
    return *reinterpret_cast<const bf16x8*>(smem_g + b * STG + (g * 6 + t * 3 + pl) * REC +
                                            lane * 16);
  };
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                         0.f};

  int buf = 0;
  Frag6 fc;
#pragma unroll
  for (int q = 0; q < 6; ++q) fc.p[RT[q]][RP[q]] = read_frag(0, 0, RT[q], RP[q]);
  // One chunk of 64 hidden units = 8 sub-stages.  Two straight-line copies: the steady state
  // (every stage that follows exists: constant wait counts, unconditional DMA issue) and the
  // block's LAST chunk (static tail counts).  Conditional branches on `last` inside the body
  // cost ~15 us per launch in the MFMA-only variant (r03w): with one wave per SIMD an
  // instruction-fetch bubble behind a branch drains the matrix pipe.
  if constexpr ((VAR & 8192) != 0) clk[10] = __builtin_readcyclecounter();
  auto chunk = [&](auto last_tag, int c) {
    constexpr bool last = decltype(last_tag)::value;
    f32x4 bq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sub = 0; sub < 8; ++sub) {
      if constexpr ((VAR & 8192) != 0 && !last) {
        __builtin_amdgcn_sched_barrier(0);
        clk[sub] = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        Frag6 fn;
        // the piece of the hidden split that rides on this MFMA group: tile 0 under the
        // phase-A stages of tile 1 (sub 2, 3), tile 1 under the first two phase-B stages
        const bool do_piece = sub >= 2 && sub < 6;
        const int pnt = sub >= 4 ? 1 : 0;
        const int pi = ((sub & 1) << 2) + g;           // piece 0..7 of the tile
        if (do_piece && (pi & 1) == 0)
          bq = *reinterpret_cast<const f32x4*>(b1s + c * 64 + pnt * 32 + 8 * (pi >> 1) + 4 * hi);
#pragma unroll
        for (int k = 0; k < 12; ++k) {
          const int q = k >> 1, tl = k & 1;       // plane product, operand tile of the group
          if ((VAR & 1) || ((VAR & 65536) && q >= 3)) {     // (65536: three of the six products)
            asm volatile("" ::"v"(fc.p[tl][PW[q]]));
          } else if (sub < 4) {
            const int nt = sub >> 1, ks = (sub & 1) * 8 + 2 * g + tl;
            const bool first = (sub & 1) == 0 && g == 0 && q == 0;
            f32x16& H = tl == 0 ? HE[nt] : HO[nt];
            H = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fc.p[tl][PW[q]], X[ks][PX[q]],
                                                       first ? zero16 : H, 0, 0, 0);
          } else {
            const bf16x8 hb = __builtin_bit_cast(bf16x8, Hp[sub - 4][PX[q]]);
            Y[2 * g + tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fc.p[tl][PW[q]], hb,
                                                                    Y[2 * g + tl], 0, 0, 0);
          }
          const int st = sub / HALF;                    // stage of the chunk
          const bool sync_sub = sub % HALF == HALF - 1;   // last sub-stage of its stage
          const int gs = (sub % HALF) * 4 + g;          // group inside the stage
          if (g == 3 && k == 5 && sync_sub) {
            // this wave has read all of stage t; its own pieces of stage t + 1 have landed (the
            // younger stages stay in flight), the barrier publishes t + 1 and frees t's buffer
            __builtin_amdgcn_sched_barrier(0);
            if (!(VAR & 8)) {
              if (!(VAR & 1024)) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int tail = SPC - 2 - st < 0 ? 0 : (SPC - 2 - st > RING - 2 ? RING - 2 : SPC - 2 - st);
                if (last) wait_vm_stages(tail * HALF); else wait_vm_stages((RING - 2) * HALF);
              }
              if (!(VAR & 2048)) __builtin_amdgcn_s_barrier();
            }
            if (!(VAR & 2) && !(VAR & 16384)) {
              if (st + RING < SPC) issue(c, st + RING, buf);
              else if (!last) issue(c + 1, st + RING - SPC, buf);
            }
            buf = buf + 1 == RING ? 0 : buf + 1;
            __builtin_amdgcn_sched_barrier(0);
          }
          // VAR & 16384: the DMA of stage st + RING - 1 (into the buffer the previous barrier
          // freed) rides on the MFMAs of stage st, one 1-KB piece per six MFMAs -- twelve
          // buffer_load ... lds in a row right behind the barrier keep all four waves in the
          // issue queue of the CU's one address unit for ~700 cycles (r03x clock stamps)
          if constexpr ((VAR & 16384) != 0 && !(VAR & 2)) {
            static_assert(!(VAR & 16384) || HALF == 2, "spread DMA: 48-record stages");
            // (piece groups: the two lightest slots of the twelve)
            const int ka = do_piece ? 9 : 2, kb = do_piece ? 11 : 8;
            if ((k == ka || k == kb) && gs < 6) {
              const int j12 = gs * 2 + (k == kb ? 1 : 0);
              const int tst = st + RING - 1;
              const int bufp = buf == 0 ? RING - 1 : buf - 1;
              if (tst < SPC) issue_one(c, tst * HALF + j12 / 6, bufp, j12 % 6);
              else if (!last) issue_one(c + 1, (tst - SPC) * HALF + j12 / 6, bufp, j12 % 6);
            }
          }
          // fragment reads of the next group: one per MFMA pair; in the last group of a stage
          // all six behind the barrier, one per MFMA
          if (VAR & 64) {
            fn = fc;
          } else if (g < 3 || !sync_sub) {
            if (VAR & 4096) {           // measurement: the six reads of the next group as one burst
              if (k == 0)
#pragma unroll
                for (int i = 0; i < 6; ++i) fn.p[RT[i]][RP[i]] = read_frag(buf, gs + 1, RT[i], RP[i]);
            } else if (tl == 1) {
              fn.p[RT[q]][RP[q]] = read_frag(buf, gs + 1, RT[q], RP[q]);
            }
          } else if (k >= 6 && (sub < 7 || !last)) {
            fn.p[RT[k - 6]][RP[k - 6]] = read_frag(buf, 0, RT[k - 6], RP[k - 6]);
          }
          if (do_piece) {
            if (VAR & 4) {
              // measurement: the accumulators are consumed, the arithmetic is not done
              if (k == 11) {
                const int r = 2 * pi, kb = 2 * pnt + (r >> 3), d = (r & 7) >> 1;
                Hp[kb][0][d] = __builtin_bit_cast(unsigned, HE[pnt][r]) & 0x3f803f80u;
                Hp[kb][1][d] = __builtin_bit_cast(unsigned, HO[pnt][r]) & 0x3f803f80u;
                Hp[kb][2][d] = __builtin_bit_cast(unsigned, HE[pnt][r + 1]) & 0x3f803f80u;
              }
            } else {
              piece_slice(pnt, pi, k, bq);
            }
          }
          // pin the issue order: hipcc otherwise sinks the reads next to their use and runs a
          // piece's VALU work as one block -- with one wave per SIMD nobody else keeps the
          // matrix pipe busy meanwhile
          __builtin_amdgcn_sched_barrier(0);
        }
        fc = fn;
      }
    }
    if constexpr ((VAR & 8192) != 0 && !last) clk[8] = __builtin_readcyclecounter();
  };
  for (int c = 0; c + 1 < NC; ++c) chunk(std::false_type{}, c);
  chunk(std::true_type{}, NC - 1);

  if constexpr ((VAR & 8192) != 0) clk[11] = __builtin_readcyclecounter();
  // ---- epilogue: lane = row of X, registers = d 32 i + 8 g + 4 hi + e ------------------------
  const int row = tm * 128 + wave * 32 + li;
  if constexpr (XT) {
    // A store with lane = row puts 32 bytes into each of 32 lines per instruction and the four
    // waves' 128 of them queue up in the CU's one address unit (r03x stamps: 2.5 k cycles per
    // wave, one after the other).  The ring is free now: the wave's 32 x 256 tile goes through a
    // private patch (row stride 1040 bytes: conflict-free both ways) and leaves as whole rows.
    __syncthreads();                                    // every wave has left the last stage
    char* patch = smem_g + wave * (32 * 1040);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(patch + li * 1040 + (i * 32 + 8 * g + 4 * hi) * 4) =
            f32x4{Y[i][4 * g], Y[i][4 * g + 1], Y[i][4 * g + 2], Y[i][4 * g + 3]};
    float* P0 = p.P + ((int64_t)slice * p.M + tm * 128 + wave * 32) * XD + lane * 4;
    const int nrow = min(32, p.M - (tm * 128 + wave * 32));
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(patch + r * 1040 + lane * 16);
      if (r < nrow) *reinterpret_cast<f32x4*>(P0 + (int64_t)r * XD) = v;
    }
  } else if (row < p.M) {
    float* P = p.P + ((int64_t)slice * p.M + row) * XD;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g)
      {
        const f32x4 v = f32x4{Y[i][4 * g], Y[i][4 * g + 1], Y[i][4 * g + 2], Y[i][4 * g + 3]};
        float* o = P + i * 32 + 8 * g + 4 * hi;
        if (VAR & 128) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(o), "v"(v) : "memory");
        else if (VAR & 256) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(o), "v"(v) : "memory");
        else *reinterpret_cast<f32x4*>(o) = v;
      }
  }
  if constexpr ((VAR & 8192) != 0) {
    // entries 0..8: sub-stage boundaries of the last steady-state chunk; 9 kernel entry, 10 loop
    // start, 11 loop end, 12 / 13 the 100-MHz real-time counter at entry / after the stores landed
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    clk[13] = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t_end = __builtin_readcyclecounter();
    if (blockIdx.x == gridDim.x / 2 && lane == 0) {
#pragma unroll
      for (int i = 0; i < 20; ++i) g_x6f_clk[wave][i] = clk[i];
      g_x6f_clk[wave][20] = t_end;
    }
  }
}

template <int ACT, int RING, int VAR = 0>
int launch_x6f(const FfnX6Args& a, hipStream_t s) {
  const int tiles_m = cdiv(a.M, 128);
  const int NC = a.F / a.S / 64;
  const size_t lds = (size_t)RING * XSTAGE * ((VAR & 512) ? 2 : 1) + (size_t)NC * 64 * sizeof(float);
  const int pairmap = a.S >= 2 && a.S <= 8;   // two hidden slices per XCD: X rows fetched twice, not S times
  const int grid = pairmap      ? cdiv(tiles_m, 16 / a.S) * 16
                   : a.S <= 8 ? cdiv(tiles_m, 8 / a.S) * 8
                              : tiles_m * a.S;
  auto kern = ffn_x6f_kernel<ACT, RING, VAR>;
  WN_MAX_DYN_LDS(kern, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a, tiles_m, pairmap);
  WN_HIP(hipGetLastError());
  return 0;
}

}  // namespace


int ffn_x6f_clocks(unsigned long long* out) {
  WN_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_x6f_clk), sizeof(g_x6f_clk)));
  return 0;
}

int x6_split_perm(const float* src, int R, int K, int ld, void* dst, hipStream_t s) {
  WN_CHECK(src && dst && R > 0 && K > 0 && K % 16 == 0 && ld % 4 == 0, "x6_split_perm: shape");
  const int64_t n = (int64_t)cdiv(R, 32) * (K / 16) * 64;
  hipLaunchKernelGGL(x6_split_perm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                     src, R, K, ld, reinterpret_cast<char*>(dst));
  WN_HIP(hipGetLastError());
  return 0;
}

// hidden slices: the largest S in {1, 2, 4, .., 32} with tiles_m x S blocks on the 256 CUs
// (one block per CU) and whole 64-unit chunks per slice.  S = 32 (one 64-unit chunk per block at
// F = 2048) is the small-batch end: a single utterance (M ~ 300 rows = 3 row tiles) then runs
// on 96 CUs for prologue + ONE chunk instead of on 48 for two (round 5: 23 -> 17 us per module
// at B = 1; the v_mfma_f32 pair it replaced there took 58 us).
int ffn_x6f_split(int M, int F) {
  const int tiles_m = cdiv(M, 128);
  int S = 1;
  while (S < 32 && tiles_m * (S * 2) <= 256 && F % (S * 2 * 64) == 0) S *= 2;
  return S;
}

bool ffn_x6f_supported(int M, int D, int F, int act) {
  if (D != XD || M <= 0 || F <= 0 || F % 64 != 0) return false;
  if (act != ACT_SILU && act != ACT_RELU) return false;   // (GELU: exact erf spills here)
  const int S = ffn_x6f_split(M, F);
  if (F % (S * 64) != 0 || F / S > 2048) return false;
  if ((int64_t)cdiv(M, 32) * X3_TILE * XKB >= ((int64_t)1 << 40)) return false;
  // (round 4 kept batches that fill less than half the CUs on the GEMM pair / the v_mfma_f32
  // kernels; measured in round 5 the fused kernel wins at every size -- B = 1: 3.80 -> 3.08 ms
  // per decode, B = 2: 3.79 -> 3.14 -- because what such a batch pays for is launches and
  // dependent phases, not matrix-pipe time.  ffn_x6f = 3 restores the old rule for the A/B.)
  return tune().ffn_x6f != 3 || cdiv(M, 128) * S >= 128;
}

int ffn_x6f(const FfnX6Args& a, hipStream_t s) {
  WN_CHECK((a.X3 || (a.X && a.ldx % 4 == 0)) && a.W13 && a.W2p && a.b1 && a.P && a.M > 0 && a.D == XD && a.S > 0 &&
               a.F % (a.S * 64) == 0 && a.F / a.S <= 2048 && (a.S & (a.S - 1)) == 0 && a.S <= 32,
           "ffn_x6f: bad arguments");
#define WN_X6F(RING, VAR)                                            \
  switch (a.act) {                                                   \
    case ACT_SILU: return launch_x6f<ACT_SILU, RING, VAR>(a, s);     \
    case ACT_RELU: return launch_x6f<ACT_RELU, RING, VAR>(a, s);     \
    default: break;                                                  \
  }
  // the default kernel with shader-clock stamps (same results; bench.py samples the clock the
  // kernel ran at through it, tools/ffn_clocks.py)
  if (tune().ffn_x6f_var == 25088 && a.act == ACT_SILU) {
    if (a.X3) return launch_x6f<ACT_SILU, 3, 25088 + 32768>(a, s);
    return launch_x6f<ACT_SILU, 3, 25088>(a, s);
  }
#ifdef WN_ABLATION
  // measurement builds only (python -m wenet_amd.build with WN_ABLATION=1): variants that leave
  // out a part of the kernel -- WRONG RESULTS BY DESIGN -- and the older stage shapes
  if (tune().ffn_x6f_var != 0 && a.act == ACT_SILU) {
    switch (tune().ffn_x6f_var) {
      case 1: return launch_x6f<ACT_SILU, 6, 1>(a, s);
      case 2: return launch_x6f<ACT_SILU, 6, 2>(a, s);
      case 4: return launch_x6f<ACT_SILU, 6, 4>(a, s);
      case 8: return launch_x6f<ACT_SILU, 6, 8>(a, s);
      case 64: return launch_x6f<ACT_SILU, 6, 64>(a, s);
      case 10: return launch_x6f<ACT_SILU, 6, 10>(a, s);
      case 78: return launch_x6f<ACT_SILU, 6, 78>(a, s);
      case 74: return launch_x6f<ACT_SILU, 6, 74>(a, s);
      case 14: return launch_x6f<ACT_SILU, 6, 14>(a, s);
      case 76: return launch_x6f<ACT_SILU, 6, 76>(a, s);
      case 70: return launch_x6f<ACT_SILU, 6, 70>(a, s);
      case 1094: return launch_x6f<ACT_SILU, 6, 1094>(a, s);
      case 4096: return launch_x6f<ACT_SILU, 6, 4096>(a, s);
      case 4110: return launch_x6f<ACT_SILU, 6, 4110>(a, s);
      case 2118: return launch_x6f<ACT_SILU, 6, 2118>(a, s);
      case 128: return launch_x6f<ACT_SILU, 6, 128>(a, s);
      case 8704: return launch_x6f<ACT_SILU, 3, 8704>(a, s);     // burst-DMA kernel + clock stamps
      case 82432: return launch_x6f<ACT_SILU, 3, 82432>(a, s);   // measurement: half of the MFMAs
      case 90624: return launch_x6f<ACT_SILU, 3, 90624>(a, s);   // ... + clock stamps
      case 512: return launch_x6f<ACT_SILU, 3, 512>(a, s);       // DMA of a stage as one burst behind the barrier
      case 8768: return launch_x6f<ACT_SILU, 3, 8768>(a, s);     // ... without fragment reads
      case 8708: return launch_x6f<ACT_SILU, 3, 8708>(a, s);     // ... without pieces
      case 8706: return launch_x6f<ACT_SILU, 3, 8706>(a, s);     // ... without DMA
      case 8782: return launch_x6f<ACT_SILU, 3, 8782>(a, s);     // ... MFMAs only
      default: break;
    }
  }
  if (tune().ffn_x6f_ring == 4) { WN_X6F(4, 0) }
  else if (tune().ffn_x6f_ring == 5) { WN_X6F(5, 0) }
  else if (tune().ffn_x6f_ring >= 6) { WN_X6F(6, 0) }
#endif
  // three stages of 48 records; X as its plane image where the producer wrote one
  if (a.X3) { WN_X6F(3, 16896 + 32768) }
  WN_X6F(3, 16896)
#undef WN_X6F
  set_error("ffn_x6f: unsupported activation");
  return -1;
}

}  // namespace wn
