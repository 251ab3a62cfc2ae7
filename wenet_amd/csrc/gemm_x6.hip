// fp32 GEMM on the bf16 matrix cores: C = act(A W^T + bias) with every fp32 operand
// carried as THREE bf16 planes whose sum is the fp32 value exactly,
//     x = x0 + x1 + x2,  x0 = bf16(x), x1 = bf16(x - x0), x2 = x - x0 - x1
// (8 + 8 + 8 significand bits; both subtractions and the last conversion are exact for
// |x| >= 2^-108, below that the sum is off by less than 2^-133: csrc/x6.h), and
// the product formed from six of the nine plane products, accumulated in fp32:
//     a b ~ a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0,
// each of them EXACT in fp32 (8 x 8 bits).  Dropped: a1 b2 + a2 b1 + a2 b2 <= 2^-26 |a b|,
// a quarter of the half-ulp an fp32 multiply-add rounds away itself -- the result is an
// fp32 GEMM with a different (and not larger) rounding error, not a reduced-precision
// one; tests/test_gpu_x6.py holds it against fp64 next to the v_mfma_f32 kernel.
//
// Why: gfx950 runs v_mfma_f32_32x32x16_bf16 at 16x the rate of v_mfma_f32_32x32x2_f32
// (2.5 PFLOP/s against 157 TFLOP/s dense), so six bf16 products cost 6/16 of the one fp32
// product -- and they share their fragments: 3 + 3 plane fragments feed 6 MFMAs where a
// plain bf16 GEMM reads 1 + 1 for one, which halves the LDS and DMA bytes per MFMA.
//
// Operand image ("X3", x6_split_kernel or a producer's epilogue): for a matrix of R rows
// and K columns, records of 1 KB indexed [K/16][ceil(R/32)][plane] holding 32 rows x 16 k
// of one plane as [k half h][row][8 bf16] -- byte h*512 + row*16 -- which is exactly the
// order in which the 64 lanes of a wave read a 32x32x16 MFMA operand (lane = 32 h + row):
// the DMA global -> LDS is linear (64 lanes x 16 B = one record), the ds_read_b128 of a
// fragment is linear (no bank conflicts, no swizzle), and a whole stage of a block tile
// (all row tiles x planes of one k block) is ONE contiguous piece of the image.
//
// Kernel: BM (128 | 256) x 256 block tile, 8 waves as 2 (M) x 4 (N), wave tile BM/2 x 64,
// one k block (16) per stage: BM/32*3 + 24 records = 36 / 48 KB, ring of 4 / 3 stages, all
// but the one being read in flight.  Per stage ONE barrier, late in the stage's MFMAs (see
// the loop).  MFMA operands are
// swapped (the W fragment is the "A" of the instruction): a lane then owns one ROW of C
// and 4 consecutive columns per register quad, so fp32 C is stored in 16-B pieces and an
// X3 image of C (EPI 2: the next GEMM's operand, e.g. the FFN hidden tensor) in whole
// 16-B plane pieces after one half-wave exchange (v_permlane32_swap).
#include "common.h"
#include "gemm_epilogue.h"
#include "kernels.h"
#include "x6.h"

namespace wn {

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

constexpr int REC = X3_REC;        // one (k block, 32-row tile, plane) record
constexpr int TILE3 = X3_TILE;     // the three planes of a tile and k block
constexpr int XBN = 256;

// fp32 [R][ld] (K columns) -> X3 image.  One thread per 16-B piece and plane triple:
// 32 consecutive threads = the 32 rows of a tile (512 contiguous bytes per plane).
__global__ __launch_bounds__(256) void x6_split_kernel(const float* __restrict__ src, int R,
                                                       int K, int ld, char* __restrict__ dst) {
  const int tiles = (R + 31) >> 5, nkb = K >> 4;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int row_l = (int)(idx & 31), h = (int)((idx >> 5) & 1);
  const int64_t rest = idx >> 6;
  const int kb = (int)(rest % nkb), tile = (int)(rest / nkb);
  if (tile >= tiles) return;
  const int row = tile * 32 + row_l;
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
  if (row < R) {
    const float* s = src + (int64_t)row * ld + kb * 16 + h * 8;
    a = *reinterpret_cast<const f32x4*>(s);
    b = *reinterpret_cast<const f32x4*>(s + 4);
  }
  bf16x8 p0, p1, p2;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const Split3 sa = split3(a[e]), sb = split3(b[e]);
    p0[e] = sa.h0; p1[e] = sa.h1; p2[e] = sa.h2;
    p0[4 + e] = sb.h0; p1[4 + e] = sb.h1; p2[4 + e] = sb.h2;
  }
  char* o = dst + ((int64_t)kb * tiles + tile) * TILE3 + h * 512 + row_l * 16;
  *reinterpret_cast<bf16x8*>(o) = p0;
  *reinterpret_cast<bf16x8*>(o + REC) = p1;
  *reinterpret_cast<bf16x8*>(o + 2 * REC) = p2;
}

// EPI 0: C = resid + alpha act(acc + bias) (fp32, [M][ldc]); EPI 1: K-slice partial
// P[slice][M][N] = acc; EPI 2: X3 image of act(acc + bias) (rows M, columns N)
// CONV: the A operand is gathered (implicit GEMM of a strided convolution over a
// channels-last tensor whose X3 image has one row per input pixel): GEMM row r reads pixel
// a_pix[r] + tap_delta[tap] for the k blocks of tap = kb / conv_kbc -- the DMA addresses
// are per lane instead of linear, the LDS side is unchanged.
// AF32 (opt-in, tune().x6_af32: measured slower than plane images, see below): the A operand is
// a plain row-major fp32 matrix (p.A, p.lda) -- or, with CONV, the
// channels-last fp32 tensor itself -- and is split into its three planes IN REGISTERS after
// the fragment read: no plane image of an activation ever exists (none is written by a
// producer, none is read back), at the price of ~50 VALU operations per A fragment.  The A
// part of a stage is then 32 rows x 64 B per tile, fetched by 4 lanes per row (16 cache
// lines per DMA instruction instead of 8) into row-major LDS rows whose four 16-B slots are
// XOR-swizzled with (row >> 2) & 3 on the source side, so that the two ds_read_b128 of a
// lane's 8 k values are conflict-free.
// NW = 4: a 128-row tile run by FOUR waves side by side along N (each owns all four A tiles:
// the 8-wave 256-row kernel's lower half), ring of 2 stages = 72 KB, <= 256 registers: two
// blocks share a CU.
// probe & 4 (measurement): shader-clock stamps of block gridDim.x / 2: per wave entry, loop
// start, loop end, kernel end and the 100-MHz real-time counter at entry / end
__device__ unsigned long long g_x6_clk[8][8];

template <int BM, int EPI, int ACT, bool CONV = false, bool AF32 = false, int NW = 8>
__global__ __launch_bounds__(NW * 64, 2) void gemm_x6_kernel(X6Args p, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem_x[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  unsigned long long ck0 = 0, ck1 = 0, ck2 = 0, rt0 = 0;
  if (p.probe & 4) { ck0 = __builtin_readcyclecounter(); rt0 = __builtin_amdgcn_s_memrealtime(); }
  static_assert(NW == 8 || (NW == 4 && BM == 128), "4 waves: 128-row tiles");
  constexpr int TA = BM / 32 / (NW / 4);       // A tiles (32 rows) per wave
  constexpr int A_TILE = AF32 ? 2048 : TILE3;  // bytes of a 32-row A tile and k block
  constexpr int A_BYTES = (BM / 32) * A_TILE;
  constexpr int STAGE = A_BYTES + 8 * TILE3;   // 36 / 48 KB (AF32: 32 / 40 KB)
  constexpr int NP = STAGE / REC;              // DMA pieces per stage
  constexpr int RING = NW == 4 ? 2 : BM == 128 ? 4 : 3;   // stages in LDS (72 / 144 / 144 KB)

  const int nblk = tiles_m * tiles_n;
  const int bid = xcd_block_order(blockIdx.x, nblk * p.ksplit);
  const int slice = bid / nblk, tb = bid - slice * nblk;
  // M fastest inside groups of 4 M panels (gemm_bf16p.hip): the blocks an XCD runs at a
  // time share few A and W panels
  constexpr int GM = 4;
  const int per_group = GM * tiles_n;
  const int grp = tb / per_group, in_grp = tb - grp * per_group;
  const int gm = min(GM, tiles_m - grp * GM);
  const int tm = grp * GM + in_grp % gm, tn = in_grp / gm;
  const int m0 = p.row0 + tm * BM, n0 = tn * XBN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < NW);
  const int wm = wave >> 2, wn = wave & 3;
  const int hi = lane >> 5, li = lane & 31;

  const int Ta = CONV ? p.a_tiles : (p.M + 31) >> 5, Tb = (p.N + 31) >> 5;
  const int nkb_all = p.K >> 4;
  const int nkb = nkb_all / p.ksplit, kb0 = slice * nkb;
  // Buffer descriptors address 2 GB at most.  fp32 A (AF32): the whole matrix; plane images:
  // ONE k-block slab (all row tiles x planes of a k block: rows / 32 * 3 KB) per descriptor,
  // rebuilt per stage from a 64-bit base -- an image may have any size (conv1's at config 3:
  // 3.9 GB), only a slab must stay under 2 GB (22 M rows).
  const int64_t slab_a = (int64_t)Ta * TILE3, slab_b = (int64_t)Tb * TILE3;
  const __amdgpu_buffer_rsrc_t ra32 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.A), 0, (int)min(p.a_bytes, (int64_t)0x7fffffff), 0x00020000);
  // Stage g = k block kb0 + g: A records of the block's BM/32 row tiles (contiguous in the
  // image), then the 8 W tiles.  Piece j of the stage goes to wave j % 8.  Tiles past the
  // last one read the next k block's records or (buffer bounds) zeros: they only reach
  // rows / columns of C that are never stored.
  const unsigned vlane = (unsigned)lane * 16u;
  // CONV: base pixel of this lane's row in each A piece the wave issues (planes: piece j
  // -> row tile j / 3, lane -> row lane & 31; AF32: piece j -> rows (j / 2) * 32 + (j & 1) *
  // 16 + lane / 4).  AF32 without CONV: the lane's byte offset into A instead.
  constexpr int NPA = (A_BYTES / REC + NW - 1) / NW;
  int pix[NPA];
  if (CONV || AF32) {
#pragma unroll
    for (int q = 0; q < NPA; ++q) {
      const int j = q * NW + wave;
      const int rl = AF32 ? (j >> 1) * 32 + (j & 1) * 16 + (lane >> 2) : (j / 3) * 32 + li;
      const int row = min(m0 + rl, p.M - 1);
      if (CONV) pix[q] = p.a_pix[row];
      else pix[q] = row * p.lda * 4 + (((lane & 3) ^ ((rl >> 2) & 3)) << 4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // before the DMA ring starts
#pragma unroll
    for (int q = 0; q < NPA; ++q) asm volatile("" : "+v"(pix[q]));
  }
  // DMA of stage g = k block kb0 + g: the stage's descriptors / offsets (scalar), then its
  // pieces one by one -- piece q of this wave is record j = q NW + wave of the stage.
  // (the descriptor pair lives in plain locals: a struct with buffer-resource members does not
  // survive the host pass)
  __amdgpu_buffer_rsrc_t d_ra = ra32, d_rb = ra32;
  int d_sa = 0, d_sb = 0, d_delta = 0;
  char* d_dst = smem_x;
  auto make_desc = [&](int g) {
    const int kb = kb0 + g;
    d_dst = smem_x + (g % RING) * STAGE;
    d_sa = (m0 >> 5) * TILE3; d_delta = 0;
    int ka = kb, kw = kb;
    if (CONV) {
      // K is walked channel block by channel block with the taps inside (p.conv_taps > 0):
      // the 9 taps of a 16-channel block re-read the same input pixels, and the block tiles
      // an XCD runs together then keep that working set in its L2 (28 frames x 39 pixels x
      // 96 B per tile) instead of fetching every tap from HBM again.  The weight image
      // stays tap-major: its k block is tap * conv_kbc + channel block.
      int tap = kb / p.conv_kbc;
      ka = kb - tap * p.conv_kbc;
      if (p.conv_taps > 0) { ka = kb / p.conv_taps; tap = kb - ka * p.conv_taps; }
      kw = tap * p.conv_kbc + ka;
      d_sa = AF32 ? ka * 64 : 0;
      d_delta = p.tap_delta[tap];
    } else if (AF32) {
      d_sa = kb * 64;
    }
    if (!AF32)
      d_ra = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(p.A3)) + ka * slab_a, 0,
          (int)min(slab_a, (int64_t)0x7fffffff), 0x00020000);
    d_rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.B3)) + kw * slab_b, 0,
        (int)min(slab_b, (int64_t)0x7fffffff), 0x00020000);
    d_sb = (n0 >> 5) * TILE3;
  };
  constexpr int NPW = (NP + NW - 1) / NW;          // pieces per wave and stage (at most)
  auto issue_piece = [&](int q) {
#ifdef WN_ABLATION
    if (p.probe & 2) return;                       // ablation: no DMA (tools/bench_x6.py)
#endif
    const int j = q * NW + wave;
    const int qa = q < NPA ? q : NPA - 1;          // (A pieces only: j < A_BYTES / REC)
    if (j < A_BYTES / REC) {
      if (AF32) {
        unsigned vo = (unsigned)pix[qa];
        if (CONV) {
          const int rl = (j & 1) * 16 + (lane >> 2);
          vo = (unsigned)((pix[qa] + d_delta) * p.conv_kbc * 64 +
                          (((lane & 3) ^ ((rl >> 2) & 3)) << 4));
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(d_ra, (lds_ptr)(d_dst + j * REC), 16, vo, d_sa, 0,
                                                 0);
      } else if (CONV) {
        const int P = pix[qa] + d_delta;
        const unsigned vo = (unsigned)(((P >> 5) * 3 + j % 3) * REC + hi * 512 + (P & 31) * 16);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(d_ra, (lds_ptr)(d_dst + j * REC), 16, vo, d_sa, 0,
                                                 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(d_ra, (lds_ptr)(d_dst + j * REC), 16, vlane,
                                                 d_sa + j * REC, 0, 0);
      }
    } else if (j < NP)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(d_rb, (lds_ptr)(d_dst + j * REC), 16, vlane,
                                               d_sb + (j - A_BYTES / REC) * REC, 0, 0);
  };
  auto issue = [&](int g) {
    make_desc(g);
#pragma unroll
    for (int q = 0; q < NPW; ++q) issue_piece(q);
  };

  struct FA { bf16x8 p[3]; };
  struct FB { bf16x8 p[2][3]; };
  auto loadA = [&](int g, int i) {
    FA f;
    if constexpr (AF32) {
      // row li of the tile: 64 B, slots swizzled; this lane's k half = slots 2 hi, 2 hi + 1
      const char* st = smem_x + (g % RING) * STAGE + (wm * TA + i) * A_TILE + li * 64;
      const int fz = (li >> 2) & 3;
      const f32x4 u = *reinterpret_cast<const f32x4*>(st + (((2 * hi) ^ fz) << 4));
      const f32x4 v = *reinterpret_cast<const f32x4*>(st + (((2 * hi + 1) ^ fz) << 4));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const Split3 su = split3(u[e]), sv = split3(v[e]);
        f.p[0][e] = su.h0; f.p[1][e] = su.h1; f.p[2][e] = su.h2;
        f.p[0][4 + e] = sv.h0; f.p[1][4 + e] = sv.h1; f.p[2][4 + e] = sv.h2;
      }
    } else {
      const char* st = smem_x + (g % RING) * STAGE + (wm * TA + i) * TILE3 + lane * 16;
#pragma unroll
      for (int q = 0; q < 3; ++q) f.p[q] = *reinterpret_cast<const bf16x8*>(st + q * REC);
    }
    return f;
  };
  auto loadB = [&](int g) {
    const char* st = smem_x + (g % RING) * STAGE + A_BYTES + (wn * 2) * TILE3 + lane * 16;
    FB f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        f.p[j][q] = *reinterpret_cast<const bf16x8*>(st + j * TILE3 + q * REC);
    return f;
  };

  f32x16 acc[TA][2];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // six plane products, the small ones first, the two accumulators of the A tile
  // alternating (no MFMA waits for its predecessor's result); W fragment = MFMA "A" (rows
  // of the instruction's result = columns of C)
  auto mma = [&](const FA& a, const FB& b, int i, auto&& between) {
    constexpr int PB[6] = {2, 0, 1, 1, 0, 0}, PA[6] = {0, 2, 1, 0, 1, 0};
#ifdef WN_ABLATION
    if (p.probe & 1) {                             // ablation: no MFMAs, keep the reads alive
      asm volatile("" :: "v"(a.p[0]), "v"(a.p[1]), "v"(a.p[2]));
#pragma unroll
      for (int j = 0; j < 2; ++j) asm volatile("" :: "v"(b.p[j][0]), "v"(b.p[j][1]), "v"(b.p[j][2]));
      return;
    }
#endif
#pragma unroll
    for (int q = 0; q < 6; ++q) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b.p[j][PB[q]], a.p[PA[q]],
                                                            acc[i][j], 0, 0, 0);
      between(q);
    }
  };

  // ---- pipeline ----------------------------------------------------------------------------
  // Ring of RING stages, RING-1 of them in flight behind the one being read (3 stages of 48
  // KB for 256-row tiles, 4 of 36 KB for 128-row tiles: the K-slice GEMMs that use those
  // stream their A operand from HBM, not from L2).  Top of stage g: stages <= g landed and
  // visible, g+1 .. g+RING-1 in flight, the W fragments and the first A tile of stage g in
  // registers.  The last A tile's fragments are read while tile TA-2 is multiplied, so
  // after those MFMAs the wave is done with stage g's buffer: it waits (counted: the
  // younger stages may stay in flight) for its own pieces of stage g+1, the ONE barrier of
  // the stage makes g+1 visible and proves everyone has left stage g -- whose buffer the
  // DMA of stage g+RING then overwrites.  The fragments of stage g+1 are read during the
  // last A tile.
  constexpr int BAR = TA - 2;
  const int npw = (NP - wave + NW - 1) / NW;      // DMA pieces this wave issues per stage
  auto wait_pieces = [&](int stages_in_flight) {  // of the stages younger than the awaited one
    switch (stages_in_flight * npw) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
      case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
      case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;   // 3 x 5
    }
  };
#pragma unroll
  for (int g = 0; g < RING; ++g)
    if (g < nkb) issue(g);
  wait_pieces(min(nkb, RING) - 1);
  __builtin_amdgcn_s_barrier();
  FB fb = loadB(0);
  FA fa = loadA(0, 0);
  if (p.probe & 4) ck1 = __builtin_readcyclecounter();
  // The DMA of stage g + RING does not follow the barrier as one burst: the CU has ONE address
  // unit, a 1-KB piece occupies it for ~16 cycles, and NP pieces issued by all waves at once
  // hold every wave in the issue queue for NP x 16 cycles per stage (ffn_x6f.hip, clock stamps
  // r03x: ~700 of ~3800 cycles).  The pieces ride on the MFMAs instead: slot 0 = the A tiles
  // behind the barrier of stage g, slots 1.. = the tiles of stage g + 1 in front of ITS
  // barrier, PP pieces per slot between the plane products -- all of them issued before the
  // counted wait of stage g + 1, so the wait counts do not change.
  constexpr int NS = TA - 1 > 0 ? TA - 1 : 1;      // tiles (slots) between two barriers' waits
  constexpr int PP = (NPW + NS - 1) / NS;          // pieces per slot
  bool dnext_ok = false;
  for (int g = 0; g < nkb; ++g) {
#pragma unroll
    for (int i = 0; i < TA; ++i) {
      FA na; FB nb;
      if (i + 1 < TA) na = loadA(g, i + 1);
      else if (g + 1 < nkb) { nb = loadB(g + 1); na = loadA(g + 1, 0); }
      const int slot = i > BAR ? i - BAR - 1 : i < BAR ? TA - 1 - BAR + i : -1;
      mma(fa, fb, i, [&](int q) {
        if (TA < 3 || slot < 0 || !dnext_ok) return;
#pragma unroll
        for (int r = 0; r < PP; ++r) {
          const int qq = PP <= 3 ? (r * 6) / PP + 1 : r;     // after which plane product
          const int pc = slot * PP + r;
          if (q == qq && pc < NPW) issue_piece(pc);
        }
      });
      if (i == BAR) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // stage g fully read
        wait_pieces(max(0, min(nkb - 2 - g, RING - 2)));
        __builtin_amdgcn_s_barrier();
        if (TA < 3) {
          if (g + RING < nkb) issue(g + RING);
        } else {
          dnext_ok = g + RING < nkb;
          if (dnext_ok) make_desc(g + RING);
        }
      }
      if (i + 1 < TA) fa = na;
      else if (g + 1 < nkb) { fa = na; fb = nb; }
    }
  }

  if (p.probe & 4) ck2 = __builtin_readcyclecounter();
  // ---- epilogue: lane = row of C, registers = columns 8 g + 4 hi + e ------------------------
  const int Tm = Ta;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int cb = n0 + (wn * 2 + j) * 32;               // first column of the 32-col tile
    // (clamped addresses under ONE uniform test: `c < N ? load : 0` per vector made every one of
    // these loads a branch with its own wait, gemm_x6r.hip r07v)
    f32x4 bias4[4] = {};
    if (EPI != 1 && p.bias) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        bias4[g] = *reinterpret_cast<const f32x4*>(p.bias + min(cb + 8 * g + 4 * hi, p.N - 4));
    }
#pragma unroll
    for (int i = 0; i < TA; ++i) {
      const int row = m0 + (wm * TA + i) * 32 + li;
      f32x4 v[4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = acc[i][j][4 * g + e] + bias4[g][e];
          if (EPI != 1) {
            if (ACT == ACT_SILU) x = silu_fast(x);
            if (ACT == ACT_RELU) x = fmaxf(x, 0.0f);
            if (ACT == ACT_GELU) x = 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
          }
          v[g][e] = x;
        }
      if constexpr (EPI == 2) {
        // planes of the lane's 16 values; after the half-wave exchange the low lane holds
        // columns 0-15 (k block 2 t), the high lane 16-31 (k block 2 t + 1) of its row
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
        const int tile_m = (m0 >> 5) + wm * TA + i;
        const int kbn = (cb >> 4) + hi;
        if (tile_m < Tm && cb + 16 * hi < p.N) {
          char* o = reinterpret_cast<char*>(p.C3) + ((int64_t)kbn * Tm + tile_m) * TILE3 + li * 16;
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            // half 0: columns 0-7 of the k block = (own g0 | partner's g0), half 1: g1
            *reinterpret_cast<i32x4*>(o + pl * REC) =
                i32x4{q[pl][0][0], q[pl][0][1], q[pl][2][0], q[pl][2][1]};
            *reinterpret_cast<i32x4*>(o + pl * REC + 512) =
                i32x4{q[pl][1][0], q[pl][1][1], q[pl][3][0], q[pl][3][1]};
          }
        }
      } else {
        if (row >= p.M) continue;
        float* crow = EPI == 1 ? p.C + ((int64_t)slice * p.M + row) * p.N
                               : p.C + (int64_t)row * p.ldc;
        f32x4 rs[4] = {};
        if (EPI == 0 && p.resid) {                 // (the row's residual quads as one batch)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            rs[g] = *reinterpret_cast<const f32x4*>(p.resid + (int64_t)row * p.ldr +
                                                    min(cb + 8 * g + 4 * hi, p.N - 4));
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = cb + 8 * g + 4 * hi;
          if (c >= p.N) continue;
          f32x4 o = v[g];
          if constexpr (EPI == 0) {
            o *= p.alpha;
            if (p.resid) o += rs[g];    // (its own block: never contracted with the multiply)
          }
          *reinterpret_cast<f32x4*>(crow + c) = o;
        }
      }
    }
  }
  if (p.probe & 4) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long ck3 = __builtin_readcyclecounter();
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == gridDim.x / 2 && lane == 0) {
      unsigned long long* o = g_x6_clk[wave & 7];
      o[0] = ck0; o[1] = ck1; o[2] = ck2; o[3] = ck3; o[4] = rt0; o[5] = rt1; o[6] = nkb;
    }
  }
}

template <int BM, int EPI, int ACT, bool CONV = false, bool AF32 = false, int NW = 8>
int launch_x6(const X6Args& a, hipStream_t s) {
  const int tiles_m = cdiv(a.M - a.row0, BM), tiles_n = cdiv(a.N, XBN);
  const size_t lds = (size_t)(NW == 4 ? 2 : BM == 128 ? 4 : 3) *
                     ((BM / 32) * (AF32 ? 2048 : TILE3) + 8 * TILE3);
  auto kern = gemm_x6_kernel<BM, EPI, ACT, CONV, AF32, NW>;
  WN_MAX_DYN_LDS(kern, lds);
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n * a.ksplit), dim3(NW * 64), lds, s, a, tiles_m,
                     tiles_n);
  WN_HIP(hipGetLastError());
  return 0;
}

template <int BM, int EPI, bool AF32 = false, int NW = 8>
int launch_x6_act(const X6Args& a, hipStream_t s) {
  if (EPI == 1) return launch_x6<BM, EPI, ACT_NONE, false, AF32, NW>(a, s);
  switch (a.act) {
    case ACT_NONE: return launch_x6<BM, EPI, ACT_NONE, false, AF32, NW>(a, s);
    case ACT_SILU: return launch_x6<BM, EPI, ACT_SILU, false, AF32, NW>(a, s);
    case ACT_RELU: return launch_x6<BM, EPI, ACT_RELU, false, AF32, NW>(a, s);
    case ACT_GELU: return launch_x6<BM, EPI, ACT_GELU, false, AF32, NW>(a, s);
    default: break;
  }
  set_error("gemm_x6: unsupported activation");
  return -1;
}

}  // namespace

// 0 (default): activations reach the kernel as plane images; 1: as plain fp32 rows split in
// registers.  Measured (r02ag): the split costs more than the plane bytes it saves -- FFN w_1
// 53.8 -> 63.7 us, w_2 54.6 -> 60.7, conv2 921 -> 1055 (+ conv1 185 -> 125), 8192 x 4096 x
// 4096 1124 -> 1312 us, decode step 7.28 -> 7.4-7.7 ms.

namespace {
// C[r][c] = relu(sum_s P[s][r][c] + bias[c]): the K-slice partials of conv2's last tiles
__global__ __launch_bounds__(256) void conv_tail_reduce_kernel(const float* __restrict__ P, int S,
                                                               int rows, int N4,
                                                               const float* __restrict__ bias,
                                                               float* __restrict__ C, int ldc) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)rows * N4) return;
  const int r = (int)(i / N4), c = (int)(i - (int64_t)r * N4) * 4;
  f32x4 acc = bias ? *reinterpret_cast<const f32x4*>(bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  for (int sl = 0; sl < S; ++sl)
    acc += *reinterpret_cast<const f32x4*>(P + ((int64_t)sl * rows + r) * (N4 * 4) + c);
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = fmaxf(acc[e], 0.0f);
  *reinterpret_cast<f32x4*>(C + (int64_t)r * ldc + c) = acc;
}
}  // namespace

int gemm_x6_clocks(unsigned long long* out) {
  WN_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_x6_clk), sizeof(g_x6_clk)));
  return 0;
}

size_t x6_bytes(int R, int K) { return (size_t)(K / 16) * cdiv(R, 32) * TILE3; }

int x6_split(const float* src, int R, int K, int ld, void* dst, hipStream_t s) {
  WN_CHECK(src && dst && R > 0 && K > 0 && K % 16 == 0 && ld % 4 == 0, "x6_split: shape");
  const int64_t n = (int64_t)cdiv(R, 32) * (K / 16) * 64;
  hipLaunchKernelGGL(x6_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, R,
                     K, ld, reinterpret_cast<char*>(dst));
  WN_HIP(hipGetLastError());
  return 0;
}

// block rows: 256-row tiles (8 waves, one block per CU) from two rounds of the 256 CUs on;
// below that 128-row tiles on four waves, two blocks per CU -- per tile as efficient as the
// 256-row ones (8192 x 4096 x 4096: 1202 vs 1185 us) and at K = 256 faster, because the
// co-resident blocks drift apart and overlap each other's prologue / store burst (FFN w_1:
// 51.9 vs 57.9 us, r02ao).  (The 8-wave form of the 128-row tile -- 63.9 us, 1400 us -- is
// kept only for the fp32-A variant.)
// From two rounds on the choice follows the LAST round: 256-row tiles cost ceil(t256 / 256)
// rounds; 128-row tiles fill a round with 512 (two per CU, together as long as one 256-row
// tile) and a remainder of up to 256 of them runs one per CU in half a round.  The CTC head of
// config 2 (31 x 17 = 527 tiles = 2 rounds + 15 tiles) pays 3 rounds as 256-row tiles, 2.5 as
// 128-row ones (144 -> ~120 us).  Ties keep the 256-row tile.
int gemm_x6_bm(int M, int N, int ksplit) {
  const int tn = cdiv(N, XBN) * ksplit;
  const int t256 = cdiv(M, 256) * tn, t128 = cdiv(M, 128) * tn;
  if (t256 < 512) return 128;
  const int rem = t128 % 512;
  const int half_rounds_128 = 2 * (t128 / 512) + (rem == 0 ? 0 : rem <= 256 ? 1 : 2);
  return half_rounds_128 < 2 * cdiv(t256, 256) ? 128 : 256;
}

int gemm_x6(const X6Args& args, hipStream_t s) {
  X6Args a = args;
  if (tune().x6_probe) a.probe = tune().x6_probe;      // ablation knob (tools/bench_x6.py --probe)
  const bool af32 = a.A != nullptr;
  WN_CHECK((a.A3 || af32) && a.B3 && a.M > 0 && a.N > 0 && a.K > 0 && a.K % 16 == 0,
           "gemm_x6: shape");
  WN_CHECK(a.ksplit >= 1 && (a.K / 16) % a.ksplit == 0, "gemm_x6: K split");
  WN_CHECK(a.N % 4 == 0 && (a.epi != 2 || a.N % 16 == 0), "gemm_x6: N");
  WN_CHECK((int64_t)cdiv(a.a_pix ? 32 * a.a_tiles : a.M, 32) * TILE3 < ((int64_t)1 << 31) &&
               (int64_t)cdiv(a.N, 32) * TILE3 < ((int64_t)1 << 31),
           "gemm_x6: a k-block slab of an operand image over 2 GB");
  WN_CHECK(!af32 || (a.a_bytes > 0 && a.a_bytes < ((int64_t)1 << 31) &&
                     (a.a_pix || a.lda % 4 == 0)), "gemm_x6: fp32 A operand");
  WN_CHECK(a.epi == 2 ? a.C3 != nullptr : a.C != nullptr, "gemm_x6: no output");
  WN_CHECK(a.epi == 1 || a.ksplit == 1, "gemm_x6: K slices need the partial epilogue");
  const int bm = a.bm ? a.bm : gemm_x6_bm(a.M, a.N, a.ksplit);
  if (a.a_pix) {
    // implicit GEMM of the subsampling conv2: fp32 C with bias + ReLU
    WN_CHECK(a.epi == 0 && a.act == ACT_RELU && a.conv_kbc > 0 && (af32 || a.a_tiles > 0) &&
                 (a.K / 16) % a.conv_kbc == 0 && a.K / 16 / a.conv_kbc <= 9,
             "gemm_x6: gathered A operand");
    // One 256-row tile per CU and round: when the last round would be less than half
    // full, its rows go to a second launch of 128-row tiles (half as long) instead --
    // 589 tiles at config 2 = 2.3 rounds become 2 rounds + 154 half tiles.
    // (rounds cut for fewer CUs, with the prefix beam search of the batch in front confined to
    // its own few CUs by a CU-masked stream, were measured in round 4: 39-44 k against 44 k
    // audio-s/s on the same box, profiles/r06b_cu_mask.txt -- removed)
    const int ncu = 256;
    const int t256 = cdiv(a.M, 256), full = t256 / ncu * ncu;
    auto run = [&](const X6Args& x, int rows) {
      if (af32) return rows == 256 ? launch_x6<256, 0, ACT_RELU, true, true>(x, s)
                                   : launch_x6<128, 0, ACT_RELU, true, true>(x, s);
      return rows == 256 ? launch_x6<256, 0, ACT_RELU, true>(x, s)
                         : launch_x6<128, 0, ACT_RELU, true, false, 4>(x, s);
    };
    // tune().x6_conv_bm = 128 (round 5, the default): the WHOLE conv2 as ONE launch of 128-row
    // tiles on four waves, two blocks per CU.  With two decodes in flight conv2 runs beside the
    // previous decode's prefix beam search, which holds 32 CUs for ~0.9 ms: the 256-row form's
    // second launch below -- the last round as K slices, 231 blocks meant for 256 free CUs --
    // then spills into a second round and doubles, its full rounds lose 12.5 %; the finer tiles
    // of one launch lose 51 us where the three launches lose ~137 (r12l: +2.3 % per step).  Alone
    // the one-launch form is 3 % slower (844 vs 822 us: -0.4 % on a plain decode()); it is the
    // default for every caller so that DecodePipeline and decode() return the same bits (the
    // K-slice tail sums in another order).  0 = the 256-row form (A/B).
    if (a.bm == 0 && tune().x6_conv_bm == 128 && !af32) return run(a, 128);
    if (a.bm == 0 && a.N <= XBN && full > 0 && t256 - full > 0 && t256 - full <= ncu / 2) {
      X6Args main = a, rest = a;
      main.M = full * 256;
      rest.row0 = full * 256;
      if (run(main, 256) != 0) return -1;
      // round 3: the remaining t256 - full tiles as K SLICES of 256-row tiles when tiles x
      // slices still fit one round -- 77 tiles x 3 slices at config 2: a third of a round
      // (+ a 60-MB reduction) instead of 154 half-height tiles that each take a whole tile's
      // time on a wave per SIMD (207 us, r05e)
      const int rem = t256 - full, nkb = a.K / 16;
      int S = 0;
      for (int t = std::min(4, ncu / rem); t >= 2; --t)
        if (nkb % t == 0) { S = t; break; }
      const int rows = a.M - full * 256;
      if (!af32 && S >= 2 && a.part &&
          a.part_bytes >= (size_t)S * rows * a.N * sizeof(float) && a.N % 4 == 0) {
        X6Args r = a;
        r.a_pix = a.a_pix + (size_t)full * 256;
        r.M = rows; r.row0 = 0; r.epi = 1; r.ksplit = S; r.C = a.part; r.bm = 256;
        if (launch_x6<256, 1, ACT_NONE, true>(r, s) != 0) return -1;
        const int64_t n4 = (int64_t)rows * (a.N / 4);
        hipLaunchKernelGGL(conv_tail_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256),
                           0, s, a.part, S, rows, a.N / 4, a.bias,
                           a.C + (int64_t)full * 256 * a.ldc, a.ldc);
        WN_HIP(hipGetLastError());
        return 0;
      }
      return run(rest, 128);
    }
    return run(a, bm);
  }
#define WN_X6(BM, AF)                                          \
  switch (a.epi) {                                             \
    case 0: return launch_x6_act<BM, 0, AF>(a, s);             \
    case 1: return launch_x6_act<BM, 1, AF>(a, s);             \
    case 2: return launch_x6_act<BM, 2, AF>(a, s);             \
    default: break;                                            \
  }
  if (bm == 128 && !af32 && a.nw != 8) {
    // 128-row tiles on four waves, two blocks per CU
    switch (a.epi) {
      case 0: return launch_x6_act<128, 0, false, 4>(a, s);
      case 1: return launch_x6_act<128, 1, false, 4>(a, s);
      case 2: return launch_x6_act<128, 2, false, 4>(a, s);
      default: break;
    }
  } else if (af32) {
    if (bm == 256) { WN_X6(256, true) } else { WN_X6(128, true) }
  } else {
    if (bm == 256) { WN_X6(256, false) } else { WN_X6(128, false) }
  }
#undef WN_X6
  set_error("gemm_x6: unknown epilogue");
  return -1;
}

}  // namespace wn
