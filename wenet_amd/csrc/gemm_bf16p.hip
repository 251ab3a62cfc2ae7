// Pipelined low-precision GEMM for the large shapes of the bf16 / fp8 modes
// (Whisper-large widths): C = epi(A[M,K] * W[N,K]^T), operands ALREADY in their
// storage type in HBM, fp32 accumulate.
//   ET = 0  bf16 A and W (the storage form of gemm_bf16s.hip; same arithmetic as
//           gemm_bf16s_kernel up to the k order inside an MFMA), C fp32 or bf16;
//   ET = 1  OCP MXFP8: e4m3 elements with one shared E8M0 (power of two) scale per
//           32 consecutive k of a row, for BOTH operands, multiplied on
//           v_mfma_scale_f32_32x32x64_f8f6f4 -- the hardware applies the two block
//           scales, at twice the bf16 MFMA rate; C fp32, or again MXFP8 with the
//           block scales computed in the epilogue (the FFN hidden tensor, whose
//           32-column blocks are the k blocks of the next GEMM).  BASELINE.json
//           configs[4] "MFMA fp8 FFN"; reference FFN: positionwise_feed_forward.py:50-58.
//
// Machine mapping, built for one 512-thread block per CU at 256 VGPRs:
//  * 256 x 256 block tile, K tile of 128 BYTES per row (64 bf16 / 128 e4m3), 8 waves
//    as 2 (M) x 4 (N), wave tile 128 x 64 = 8 accumulator tiles of a 32 x 32 MFMA
//    (128 VGPRs).
//  * operands go global -> LDS directly (buffer_load_dwordx4 ... lds, no staging
//    registers): a K tile is four 16-KB half-tiles {A rows 0-127, A rows 128-255,
//    W rows 0-127, W rows 128-255}, two K tiles of LDS (128 KB).  The LDS image of
//    a half-tile is [128 rows][8 x 16 B]; the DMA writes lane-linear, so the bank
//    swizzle slot' = slot ^ ((row >> 1) & 7) is applied to the SOURCE slot of each
//    lane and again on the fragment reads -- ds_read_b128 of 32 consecutive rows at
//    one k-slot then touches 16 distinct 16-B slots per service group.
//  * 4 phases per K tile; phase q computes one 64 x 32 quadrant of the wave tile
//    over the whole K tile (8 bf16 / 4 fp8 MFMAs = 256 matrix-pipe cycles) after
//    reading only the fragments it is the first to need (A0+B0 | B1 | A1 | -), and
//    issues the DMA of one half-tile.  Half-tiles of K tile t+1 are issued in phases
//    4t-1 .. 4t+2 (one phase after the last fragment read of K tile t-1, whose LDS
//    they overwrite), waited for with ONE counted s_waitcnt vmcnt per K tile in
//    phase 4t+3 -- the newest half-tile stays in flight across the tile boundary --
//    and read from phase 4t+4 on.  Raw s_barrier only (a __syncthreads() would drain
//    the DMA queue).  DEEP (experiment): issue order {A lo, A hi | W lo, W hi of
//    t+2} with two half-tiles in flight at the wait.
//  * the two waves of a SIMD (waves w and w+4) run one barrier apart: while one is
//    in its MFMA segment the other issues its ds_reads / DMA (MI355X_MICROARCH.md
//    "Two waves per SIMD"); measured +9..15 % over lockstep (profiles/r02c).
//  * MX block scales: one dword per (row, K tile) = the 4 E8M0 bytes of its 4 k
//    blocks, stored K-tile-major ([K/128][rows]) so that a K tile's 256 A and 256 W
//    scale dwords are two 1-KB DMA pieces (4 B per lane) next to the operand tiles.
//  * operand-swapped MFMA (D = W_frag * A_frag^T): a lane owns ONE output row and
//    4 x 4 consecutive columns per 32 x 32 tile, so the epilogue issues 16-byte
//    (fp32 C) / 8-byte (bf16 C) stores and 16-byte residual loads, and a lane pair
//    (l, l ^ 32) owns one whole 32-column MX block of the row.
//  * GELU in the epilogue: gelu_e5 below (erfc as 2^(-z G(z)), one v_exp) instead of erff.
//  * fp32 / bf16 C leaves through wave-private LDS patches as whole 128-byte row segments
//    (epilogue comment below).
#include "common.h"
#include "gemm_epilogue.h"
#include "mxfp8.h"

namespace wn {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int PBM = 256, PBN = 256;
constexpr int KROW = 128;                     // bytes of one tile row
constexpr int HALF_BYTES = 128 * KROW;        // 16 KB
constexpr int TILE_BYTES = 4 * HALF_BYTES;    // 64 KB: A lo, A hi, W lo, W hi
constexpr int SCALE_BYTES = 2048;             // per K tile: 256 A + 256 W dwords
constexpr int EPI_PITCH = 128 + 16;           // epilogue patch: [128 rows][128 B + 16 B pad]
constexpr int EPI_WAVE = 128 * EPI_PITCH;     // 18 KB per wave, 144 KB per block

#define WN_LGKM0() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
                        __builtin_amdgcn_sched_barrier(0); } while (0)

// GELU of this kernel (round 3; the result is rounded to bf16 / e4m3 or added into an fp32 stream
// that the next GEMM rounds): x Phi(x) = max(x, 0) - 0.5 |x| erfc(|x| / sqrt 2) with
// erfc(z) = 2^(-z G(z)), G a degree-5 fit of -log2(erfc z) / z on [0, 6.5] (z clamped there:
// erfc(6.5) = 4e-20).  One v_exp, no v_rcp, 12 VALU operations against 15 + two transcendentals
// of the Abramowitz-Stegun 7.1.26 form used before (w_1 of the fp8 mode: 236 -> 198 us);
// |error| < 4.7e-6 absolute and < 3.2e-3 of the value in the negative tail down to the clamp (1.8e-3
// above x = -5.6; tests/test_gelu_form.py restates it from these constants), where the A&S form's
// 1 - erf(|z|) cancels (its bf16-rounded result differs from the exact one's more often than
// this one's: 18 % vs 13 % of a fine grid on [-9, 12]).
__device__ __forceinline__ float gelu_e5(float x) {
  const float ax = fabsf(x);
  const float z = fminf(ax * 0.70710678118654752f, 6.5f);
  float g = fmaf(-0.000131776848f, z, 0.00292019276f);
  g = fmaf(g, z, -0.0270496631f);
  g = fmaf(g, z, 0.143017159f);
  g = fmaf(g, z, 0.922645434f);
  g = fmaf(g, z, 1.62697166f);
  const float e = __builtin_amdgcn_exp2f(-z * g);
  return fmaf(-0.5f * ax, e, fmaxf(x, 0.0f));
}

// CM: 0 fp32 C, 1 bf16 C, 2 MXFP8 C (+ block scales); VAR bit 0: no wave stagger,
// bit 4: DEEP issue order (experiments, tools/bench_gemm.py --variants)
// p.probe & 4 (measurement, tools/lp_clocks.py): per wave the shader clock at entry, after the
// prologue (first tile landed), after the K loop, after the epilogue's last store was issued and
// after the stores drained, + the 100-MHz real-time counter at entry / end, of one block
__device__ unsigned long long g_lp_clk[8][8];

template <int ET, int ACT, bool RESID, int CM, int VAR = 0>
__global__ __launch_bounds__(512) void gemm_lp_kernel(
    GemmArgs p, const void* __restrict__ Wq, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem_p[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  unsigned long long ck0 = 0, ck1 = 0, ck2 = 0, rt0 = 0;
  if (p.probe & 4) { ck0 = __builtin_readcyclecounter(); rt0 = __builtin_amdgcn_s_memrealtime(); }
  constexpr bool FP8 = ET == 1;
  constexpr bool DEEP = (VAR & 16) != 0;
  constexpr int ESZ = FP8 ? 1 : 2;            // bytes per element
  constexpr int KT_ELEMS = KROW / ESZ;        // k per K tile
  constexpr int NDMA = FP8 ? 3 : 2;           // DMA pieces a wave issues in phase 3

  // ---- tile assignment: XCD-contiguous chunks, inside a chunk groups of 4 M
  // panels x all N tiles with M fastest: the 32 blocks an XCD runs at a time
  // share 4 A panels and 8 W panels ------------------------------------------
  const int nblk = tiles_m * tiles_n;
  const int bid = xcd_block_order(blockIdx.x, nblk);
  constexpr int GM = 4;
  const int per_group = GM * tiles_n;
  const int grp = bid / per_group;
  const int in_grp = bid - grp * per_group;
  const int gm = min(GM, tiles_m - grp * GM);   // M panels in this group
  const int tm = grp * GM + in_grp % gm, tn = in_grp / gm;
  const int m0 = tm * PBM, n0 = tn * PBN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn_ = wave & 3;
  const int hi = lane >> 5;

  // ---- DMA descriptors -------------------------------------------------------
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.A), 0,
      (int)min((int64_t)p.M * p.lda * ESZ, (int64_t)0x7fffffff), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(Wq), 0, (int)min((int64_t)p.N * p.K * ESZ, (int64_t)0x7fffffff),
      0x00020000);
  const int nk = p.K / KT_ELEMS;
  // MX scales: [nk][pitch] dwords; wave w < 4 moves A rows w*64.., w >= 4 W rows
  const bool sc_a = wave < 4;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned*>(sc_a ? p.a_scale : p.w_scale), 0,
      FP8 ? (int)min((int64_t)nk * (sc_a ? p.a_scale_pitch : p.w_scale_pitch) * 4,
                     (int64_t)0x7fffffff) : 0,
      0x00020000);
  const int sc_pitch = __builtin_amdgcn_readfirstlane(sc_a ? p.a_scale_pitch : p.w_scale_pitch);
  unsigned vs = 0;
  if (FP8) {
    const int r = sc_a ? min(m0 + wave * 64 + lane, p.M - 1)
                       : min(n0 + (wave - 4) * 64 + lane, p.N - 1);
    vs = (unsigned)r * 4u;
  }
  // per half-tile each thread moves 2 x 16 B: piece j covers rows (j*8 + wave)*8 ..
  // +8 of the half, lane -> (row = lane >> 3, LDS slot = lane & 7), source slot
  // = slot ^ ((row >> 1) & 7)
  unsigned va[2][2], vw[2][2];   // [half][piece] byte offsets of k tile 0
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = (j * 8 + wave) * 8 + (lane >> 3);
      const int ks = (lane & 7) ^ ((r >> 1) & 7);
      const int ar = min(m0 + h * 128 + r, p.M - 1);
      const int wr = min(n0 + h * 128 + r, p.N - 1);
      va[h][j] = ((unsigned)ar * (unsigned)p.lda) * ESZ + (unsigned)ks * 16u;
      vw[h][j] = ((unsigned)wr * (unsigned)p.K) * ESZ + (unsigned)ks * 16u;
    }

  // stage half-tile `half` (0,1 = A lo/hi; 2,3 = W lo/hi) of K tile `kt` into LDS
  // tile buffer `par`
  auto stage = [&](int kt, int half, int par) {
    const int ktc = min(kt, nk - 1);         // past the end: harmless re-load
    const int soff = ktc * KROW;
    char* dst = smem_p + par * TILE_BYTES + half * HALF_BYTES + wave * 1024;
    if (half < 2) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)dst, 16, va[half][0], soff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(dst + 8192), 16, va[half][1],
                                               soff, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)dst, 16, vw[half - 2][0], soff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(dst + 8192), 16, vw[half - 2][1],
                                               soff, 0, 0);
    }
  };
  // the block-scale dwords of K tile `kt`: one 256-B piece per wave
  auto stage_scales = [&](int kt, int par) {
    if constexpr (FP8) {
      const int ktc = min(kt, nk - 1);
      char* dst = smem_p + 2 * TILE_BYTES + par * SCALE_BYTES + wave * 256;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)dst, 4, vs, ktc * sc_pitch * 4, 0, 0);
    }
  };

  // ---- fragment addresses ------------------------------------------------------
  // a lane's fragment of a 32-row block is 4 x 16 B: unit u = 16-B slot
  //   bf16: 2 u + hi          (k-step u of the 32x32x16 MFMA, 8 k at hi * 8)
  //   fp8 : 4 (u >> 1) + 2 (u & 1) + hi   (the 32x32x64 f8f6f4 operand holds, in
  //         registers 0-3, 16 k of MX block 0 of the k-step (k = 16 hi ..), in registers
  //         4-7 16 k of block 1 (k = 32 + 16 hi ..); block b takes its scale from lane
  //         group b -- probed on hardware, tools/probes/mfma_scale_probe.hip)
  const int sw = (lane >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int slot = FP8 ? 4 * (u >> 1) + 2 * (u & 1) + hi : 2 * u + hi;
    foff[u] = (lane & 31) * KROW + ((slot ^ sw) << 4);
  }
  const char* a_base = smem_p + wm * HALF_BYTES;                       // A half = wm
  const char* b_base = smem_p + (2 + (wn_ >> 1)) * HALF_BYTES + (wn_ & 1) * (64 * KROW);
  const char* sa_base = smem_p + 2 * TILE_BYTES + (wm * 128 + (lane & 31)) * 4;
  const char* sb_base = smem_p + 2 * TILE_BYTES + 1024 +
                        ((wn_ >> 1) * 128 + (wn_ & 1) * 64 + (lane & 31)) * 4;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  i32x4 fa[2][4], fb0[4], fb1[4];
  int sca[4] = {0, 0, 0, 0}, scb[2] = {0, 0};   // block scales, byte 0 / 2 = k-step 0 / 1
  auto read_a = [&](int par, int qi) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        fa[rb][u] = *reinterpret_cast<const i32x4*>(
            a_base + par * TILE_BYTES + (qi * 64 + rb * 32) * KROW + foff[u]);
  };
  auto read_b = [&](int par, int qj, i32x4 (&fb)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      fb[u] = *reinterpret_cast<const i32x4*>(b_base + par * TILE_BYTES + qj * 32 * KROW +
                                              foff[u]);
  };
  auto read_scales = [&](int par) {
    if constexpr (FP8) {
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
        sca[mb] = *reinterpret_cast<const int*>(sa_base + par * SCALE_BYTES + mb * 128);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
        scb[nb] = *reinterpret_cast<const int*>(sb_base + par * SCALE_BYTES + nb * 128);
    }
  };
  // after the wait: lane group hi supplies the scale of MX block 2 ks + hi of its row
  // -> shift so that byte 0 / 2 hold the scales of k-step 0 / 1 (the MFMA's op_sel picks
  // a byte, uniformly per wave)
  auto align_scales = [&]() {
    if constexpr (FP8) {
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) sca[mb] = (unsigned)sca[mb] >> (8 * hi);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) scb[nb] = (unsigned)scb[nb] >> (8 * hi);
    }
  };
  auto mma = [&](int qi, int qj, const i32x4 (&fb)[4]) {
    __builtin_amdgcn_s_setprio(1);
    if constexpr (FP8) {
      i32x8 w8[2], a8[2][2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        w8[ks] = __builtin_shufflevector(fb[2 * ks], fb[2 * ks + 1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
          a8[rb][ks] = __builtin_shufflevector(fa[rb][2 * ks], fa[rb][2 * ks + 1], 0, 1, 2, 3,
                                               4, 5, 6, 7);
      }
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        acc[2 * qi + rb][qj] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
            w8[0], a8[rb][0], acc[2 * qi + rb][qj], 0, 0, 0, scb[qj], 0, sca[2 * qi + rb]);
      }
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        acc[2 * qi + rb][qj] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
            w8[1], a8[rb][1], acc[2 * qi + rb][qj], 0, 0, 2, scb[qj], 2, sca[2 * qi + rb]);
      }
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
          acc[2 * qi + rb][qj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              __builtin_bit_cast(bf16x8, fb[u]), __builtin_bit_cast(bf16x8, fa[rb][u]),
              acc[2 * qi + rb][qj], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // one K tile = 4 phases; `par` is the tile's LDS buffer (compile-time in the
  // unrolled callers)
  auto ktile = [&](int t, int par) {
    // phase 0: A rows 0-63 + W cols 0-31 (+ the tile's block scales)
    read_a(par, 0);
    read_b(par, 0, fb0);
    read_scales(par);
    stage(t + 1, DEEP ? 0 : 1, par ^ 1);
    __builtin_amdgcn_s_barrier();
    WN_LGKM0();
    align_scales();
    mma(0, 0, fb0);
    __builtin_amdgcn_s_barrier();
    // phase 1: W cols 32-63 (DEEP: the last read of this tile's W halves, retired
    // before the barrier because the next phase's DMA overwrites them)
    read_b(par, 1, fb1);
    stage(t + 1, DEEP ? 1 : 2, par ^ 1);
    if (DEEP) WN_LGKM0();
    __builtin_amdgcn_s_barrier();
    if (!DEEP) WN_LGKM0();
    mma(0, 1, fb1);
    __builtin_amdgcn_s_barrier();
    // phase 2: A rows 64-127 -- the LAST read of this tile's buffer, retired before
    // the barrier because the next phase's DMA overwrites it
    read_a(par, 1);
    if (DEEP) stage(t + 2, 2, par); else stage(t + 1, 3, par ^ 1);
    WN_LGKM0();
    __builtin_amdgcn_s_barrier();
    mma(1, 1, fb1);
    __builtin_amdgcn_s_barrier();
    // phase 3: no reads; DMA into THIS tile's buffer (+ the scales of tile t+2);
    // everything but the newest pieces must have landed before tile t+1 is read
    stage(t + 2, DEEP ? 3 : 0, par);
    stage_scales(t + 2, par);
    if constexpr (DEEP) {
      if constexpr (FP8) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      if constexpr (FP8) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    mma(1, 0, fb0);
    __builtin_amdgcn_s_barrier();
  };
  (void)NDMA;

  // ---- prologue: K tile 0 and the first pieces of K tile 1 ------------------------
  stage(0, 0, 0); stage(0, 1, 0); stage(0, 2, 0); stage(0, 3, 0);
  stage_scales(0, 0);
  if constexpr (DEEP) {
    stage(1, 2, 1); stage(1, 3, 1);
    stage_scales(1, 1);
    if constexpr (FP8) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    stage(1, 0, 1);
    stage_scales(1, 1);
    if constexpr (FP8) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (p.probe & 4) ck1 = __builtin_readcyclecounter();
  // the second wave of each SIMD runs one barrier behind the first
  if (!(VAR & 1) && wm == 1) __builtin_amdgcn_s_barrier();
  int t = 0;
  for (; t + 1 < nk; t += 2) {
    ktile(t, 0);
    ktile(t + 1, 1);
  }
  if (t < nk) ktile(t, 0);
  if (!(VAR & 1) && wm == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing (clamped) DMA
  if (p.probe & 4) ck2 = __builtin_readcyclecounter();

  // ---- epilogue: lane = output row, 4 consecutive columns per register quad ------
  const int hi4 = hi * 4;
  if constexpr (CM != 2) {
    // fp32 / bf16 C through LDS (round 3; clock stamps r05b: the direct form -- every lane its own
    // row, 16 B (fp32) / 8 B (bf16) per store -- touched 32 cache lines per store instruction and
    // took 22 k cycles for a 128-KB bf16 tile, 60 k for an fp32 tile + residual, against 44 k for
    // the whole K = 1280 loop).  Each wave turns its 128 x 64 sub-tile through a private
    // [128 rows][128 B + 16] LDS patch (fp32: one 32-column half at a time) and stores / loads
    // the residual as whole 128-byte row segments: 8 rows per instruction.
    __builtin_amdgcn_s_barrier();   // every wave's fragment reads and trailing DMA are over
    char* wbuf = smem_p + wave * EPI_WAVE;
    const int wrow = lane & 31, rrow = lane >> 3, rch = lane & 7;
    const int row0 = m0 + wm * 128;
    if constexpr (CM == 1) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int cb = n0 + wn_ * 64 + nb * 32 + hi4;
        f32x4 bias4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = cb + 8 * g;
          bias4[g] = (p.bias && c < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + c)
                                          : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            bf16x4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float x = acc[mb][nb][4 * g + e] + bias4[g][e];
              if (ACT == ACT_SILU) x = silu_fast(x);
              if (ACT == ACT_RELU) x = fmaxf(x, 0.0f);
              if (ACT == ACT_GELU) x = gelu_e5(x);
              h[e] = (__bf16)(x * p.alpha);
            }
            *reinterpret_cast<bf16x4*>(wbuf + (mb * 32 + wrow) * EPI_PITCH +
                                       (nb * 32 + 8 * g + hi4) * 2) = h;
          }
      }
      const int cc = n0 + wn_ * 64 + rch * 8;          // 8 bf16 columns = 16 B
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        i32x4 o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          o[i] = *reinterpret_cast<const i32x4*>(wbuf + (half * 64 + i * 8 + rrow) * EPI_PITCH +
                                                 rch * 16);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(o[i]));   // reads stay out of the guards
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int r = half * 64 + i * 8 + rrow;
          if (row0 + r < p.M && cc < p.N)
            *reinterpret_cast<i32x4*>(reinterpret_cast<__bf16*>(p.C) +
                                      (int64_t)(row0 + r) * p.ldc + cc) = o[i];
        }
      }
    } else {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int cb0 = n0 + wn_ * 64 + nb * 32;
        f32x4 bias4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = cb0 + hi4 + 8 * g;
          bias4[g] = (p.bias && c < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + c)
                                          : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const int cc = cb0 + rch * 4;                  // 4 fp32 columns = 16 B
        f32x4 rs[2][8];
        if constexpr (RESID) {                         // first half's residual rows: in flight
#pragma unroll                                         // while the tile turns through LDS
          for (int i = 0; i < 8; ++i) {
            const int r = i * 8 + rrow;
            rs[0][i] = (row0 + r < p.M && cc < p.N)
                           ? *reinterpret_cast<const f32x4*>(p.resid + (int64_t)(row0 + r) * p.ldr + cc)
                           : f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float x = acc[mb][nb][4 * g + e] + bias4[g][e];
              if (ACT == ACT_SILU) x = silu_fast(x);
              if (ACT == ACT_RELU) x = fmaxf(x, 0.0f);
              if (ACT == ACT_GELU) x = gelu_e5(x);
              v[e] = x * p.alpha;
            }
            *reinterpret_cast<f32x4*>(wbuf + (mb * 32 + wrow) * EPI_PITCH + (8 * g + hi4) * 4) = v;
          }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          if constexpr (RESID) {
            if (half == 0) {
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                const int r = 64 + i * 8 + rrow;
                rs[1][i] = (row0 + r < p.M && cc < p.N)
                               ? *reinterpret_cast<const f32x4*>(p.resid +
                                                                 (int64_t)(row0 + r) * p.ldr + cc)
                               : f32x4{0.f, 0.f, 0.f, 0.f};
              }
            }
          }
          f32x4 o[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            o[i] = *reinterpret_cast<const f32x4*>(wbuf + (half * 64 + i * 8 + rrow) * EPI_PITCH +
                                                   rch * 16);
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(o[i]));   // reads stay out of the guards
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int r = half * 64 + i * 8 + rrow;
            if constexpr (RESID) o[i] += rs[half][i];
            if (row0 + r < p.M && cc < p.N)
              *reinterpret_cast<f32x4*>(p.C + (int64_t)(row0 + r) * p.ldc + cc) = o[i];
          }
        }
      }
    }
  } else {
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int cb0 = n0 + wn_ * 64 + nb * 32;          // the 32-column block
    const int cb = cb0 + hi4;                         // + 8 g + (0..3)
    f32x4 bias4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = cb + 8 * g;
      bias4[g] = (p.bias && c < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + c)
                                      : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const int row = m0 + wm * 128 + mb * 32 + (lane & 31);
      f32x4 v[4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = acc[mb][nb][4 * g + e] + bias4[g][e];
          if (ACT == ACT_SILU) x = silu_fast(x);
          if (ACT == ACT_RELU) x = fmaxf(x, 0.0f);
          if (ACT == ACT_GELU) x = gelu_e5(x);
          v[g][e] = x * p.alpha;
        }
      {
        // MXFP8 C: the lane pair (l, l ^ 32) holds the 32 columns of one block
        float amax = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(v[g][e]));
        amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
        const int E = mx_e8m0(amax);
        const float inv = mx_inv_scale(E);
        int q[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) q[g] = mx_pack4(v[g][0] * inv, v[g][1] * inv,
                                                    v[g][2] * inv, v[g][3] * inv);
        // lanes < 32 own columns 8g..8g+3, lanes >= 32 columns 8g+4..8g+7: after two
        // half-swaps the low lane holds columns 0-15, the high lane 16-31 (q0,q2,q1,q3)
        {
          auto s0 = __builtin_amdgcn_permlane32_swap(q[0], q[2], false, false);
          auto s1 = __builtin_amdgcn_permlane32_swap(q[1], q[3], false, false);
          q[0] = s0[0]; q[2] = s0[1]; q[1] = s1[0]; q[3] = s1[1];
        }
        if (row < p.M && cb0 < p.N) {
          unsigned char* cq = reinterpret_cast<unsigned char*>(p.C) + (int64_t)row * p.ldc +
                              cb0 + hi * 16;
          *reinterpret_cast<i32x4*>(cq) = i32x4{q[0], q[2], q[1], q[3]};
          if (hi == 0)
            reinterpret_cast<unsigned char*>(p.c_scale)[((int64_t)(cb0 >> 7) * p.c_scale_pitch +
                                                         row) * 4 + ((cb0 >> 5) & 3)] =
                (unsigned char)E;
        }
      }
    }
  }
  }
  if (p.probe & 4) {
    const unsigned long long ck3 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long ck4 = __builtin_readcyclecounter();
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    const int want = (p.probe & 8) ? 0 : (p.probe & 16) ? (int)gridDim.x - 1 : (int)gridDim.x / 2;
    if ((int)blockIdx.x == want && lane == 0) {
      unsigned long long* o = g_lp_clk[wave & 7];
      o[0] = ck0; o[1] = ck1; o[2] = ck2; o[3] = ck3; o[4] = ck4; o[5] = rt0; o[6] = rt1;
      o[7] = (unsigned long long)nk;
    }
  }
}

template <int ET, int ACT, bool RESID, int CM, int VAR = 0>
int launch_p(const GemmArgs& a, const void* Wq, hipStream_t stream) {
  const int tiles_m = cdiv(a.M, PBM), tiles_n = cdiv(a.N, PBN);
  size_t lds = 2 * TILE_BYTES + (ET == 1 ? 2 * SCALE_BYTES : 0);
  if (CM != 2 && lds < (size_t)8 * EPI_WAVE) lds = (size_t)8 * EPI_WAVE;
  auto kern = gemm_lp_kernel<ET, ACT, RESID, CM, VAR>;
  WN_MAX_DYN_LDS(kern, lds);
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), lds, stream, a, Wq, tiles_m,
                     tiles_n);
  WN_HIP(hipGetLastError());
  return 0;
}

template <int ET>
int dispatch_p(const GemmArgs& args, const void* W, hipStream_t stream) {
  GemmArgs a = args;
  if (tune().lp_probe) a.probe = tune().lp_probe;
  const bool resid = a.resid != nullptr;
  if (a.c_mx) {
    if constexpr (ET == 1) {
      switch (a.act) {
        case ACT_NONE: return launch_p<1, ACT_NONE, false, 2>(a, W, stream);
        case ACT_SILU: return launch_p<1, ACT_SILU, false, 2>(a, W, stream);
        case ACT_GELU: return launch_p<1, ACT_GELU, false, 2>(a, W, stream);
        case ACT_RELU: return launch_p<1, ACT_RELU, false, 2>(a, W, stream);
      }
    }
    set_error("gemm(pipelined): MXFP8 C needs MXFP8 operands");
    return -1;
  }
  if (a.c_bf16) {
    if constexpr (ET == 0) {
      switch (a.act) {
        case ACT_NONE: return launch_p<0, ACT_NONE, false, 1>(a, W, stream);
        case ACT_SILU: return launch_p<0, ACT_SILU, false, 1>(a, W, stream);
        case ACT_GELU: return launch_p<0, ACT_GELU, false, 1>(a, W, stream);
        case ACT_RELU: return launch_p<0, ACT_RELU, false, 1>(a, W, stream);
      }
    }
    set_error("gemm(pipelined): bf16 C needs bf16 operands");
    return -1;
  }
  switch (a.act) {
    case ACT_NONE: return resid ? launch_p<ET, ACT_NONE, true, 0>(a, W, stream)
                                : launch_p<ET, ACT_NONE, false, 0>(a, W, stream);
    case ACT_SILU: return resid ? launch_p<ET, ACT_SILU, true, 0>(a, W, stream)
                                : launch_p<ET, ACT_SILU, false, 0>(a, W, stream);
    case ACT_GELU: return resid ? launch_p<ET, ACT_GELU, true, 0>(a, W, stream)
                                : launch_p<ET, ACT_GELU, false, 0>(a, W, stream);
    case ACT_RELU: return resid ? launch_p<ET, ACT_RELU, true, 0>(a, W, stream)
                                : launch_p<ET, ACT_RELU, false, 0>(a, W, stream);
  }
  set_error("gemm(pipelined): unsupported epilogue");
  return -1;
}

// ---- MXFP8 quantisation of an fp32 matrix (weights at set_precision time, tests) ---
// one wave per row-chunk: lane -> 4 consecutive columns, 8 lanes = one 32-column block
__global__ __launch_bounds__(256) void mx_quantize_kernel(
    const float* __restrict__ x, int ld, int rows, int K, unsigned char* __restrict__ q,
    unsigned* __restrict__ scale, int pitch) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  for (int c0 = 0; c0 < K; c0 += 256) {
    const int c = c0 + lane * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c < K) v = *reinterpret_cast<const f32x4*>(x + (int64_t)row * ld + c);
    float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
    const int E = mx_e8m0(amax);
    const float inv = mx_inv_scale(E);
    if (c < K) {
      *reinterpret_cast<int*>(q + (int64_t)row * K + c) =
          mx_pack4(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv);
      if ((lane & 7) == 0)
        reinterpret_cast<unsigned char*>(scale)[((int64_t)(c >> 7) * pitch + row) * 4 +
                                                ((c >> 5) & 3)] = (unsigned char)E;
    }
  }
}

}  // namespace

int gemm_lp_clocks(unsigned long long* out) {
  WN_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lp_clk), sizeof(g_lp_clk)));
  return 0;
}

// Shapes the pipelined kernel takes (everything else stays on gemm_bf16s_kernel)
bool gemm_bf16p_supported(const GemmArgs& a) {
  const int esz = a.fp8 ? 1 : 2;
  const int kt = 128 / esz;
  return !a.glu && a.a_row_off == nullptr && a.K % kt == 0 && a.K >= 2 * kt &&
         a.N % 8 == 0 && (a.lda * esz) % 16 == 0 &&
         (a.c_mx ? (a.N % 32 == 0 && a.ldc % 16 == 0) : a.c_bf16 ? a.ldc % 8 == 0 : a.ldc % 4 == 0) &&
         (a.resid == nullptr || a.ldr % 4 == 0) && !((a.c_bf16 || a.c_mx) && a.resid) &&
         (int64_t)a.M * a.lda * esz < (int64_t(1) << 31) &&
         (int64_t)a.N * a.K * esz < (int64_t(1) << 31);
}

int gemm_bf16_pipelined(const GemmArgs& a, const void* Wh, hipStream_t stream) {
  WN_CHECK(!a.fp8 && gemm_bf16p_supported(a), "gemm(bf16 pipelined): unsupported shape");
  return dispatch_p<0>(a, Wh, stream);
}

// A: e4m3 [M][lda] (a.A reinterpreted, lda in elements) with a.a_scale; Wq: e4m3 [N][K]
// with a.w_scale; scales as mx_quantize() lays them out.
int gemm_mxfp8(const GemmArgs& a, const void* Wq, hipStream_t stream) {
  WN_CHECK(a.fp8 && gemm_bf16p_supported(a), "gemm(mxfp8): unsupported shape");
  WN_CHECK(a.a_scale && a.w_scale && a.a_scale_pitch >= a.M && a.w_scale_pitch >= a.N,
           "gemm(mxfp8): block scales missing");
  WN_CHECK(!a.c_mx || (a.c_scale && a.c_scale_pitch >= a.M), "gemm(mxfp8): C scales missing");
  return dispatch_p<1>(a, Wq, stream);
}

int mx_quantize(const float* x, int ld, int rows, int K, void* q, unsigned* scale, int pitch,
                hipStream_t s) {
  WN_CHECK(K % 32 == 0 && ld % 4 == 0 && pitch >= rows, "mx_quantize: K % 32, ld % 4");
  hipLaunchKernelGGL(mx_quantize_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, ld, rows, K,
                     reinterpret_cast<unsigned char*>(q), scale, pitch);
  WN_HIP(hipGetLastError());
  return 0;
}

}  // namespace wn
