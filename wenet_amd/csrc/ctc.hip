// CTC tail on the GPU: log-softmax + top-k per frame, greedy collapse, and the
// CTC prefix beam search of wenet/models/transformer/search.py:127-249.
#include <math.h>

#include "kernels.h"

namespace wn {

namespace {

// ===========================================================================
// log_softmax statistics + top-k of one row per 256-thread block.
// Replaces CTC.log_softmax (ctc.py:73-81) + `logp.topk(beam_size)`
// (search.py:158) / `ctc_probs.topk(1)` (search.py:114): the (B,T',V) log-prob
// tensor is only written when the caller asks for it.
struct VI { float v; int i; };
__device__ __forceinline__ VI vi_better(VI a, VI b) {
  // larger value first; on ties the lower index
  return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ VI vi_wave(VI x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    VI y;
    y.v = __shfl_xor(x.v, o, 64);
    y.i = __shfl_xor(x.i, o, 64);
    x = vi_better(x, y);
  }
  return x;
}

__global__ __launch_bounds__(256) void ctc_row_kernel(CtcRowArgs a) {
  extern __shared__ __attribute__((aligned(16))) float srow[];
  __shared__ float red[8];
  __shared__ VI redvi[4];
  const int row = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* x = a.logits + (int64_t)row * a.ld;
  float mx = -INFINITY;
  for (int i = tid; i < a.V; i += 256) {
    float v = x[i];
    if (i == a.blank) v -= a.blank_penalty;
    srow[i] = v;
    mx = fmaxf(mx, v);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sm = 0.f;
  for (int i = tid; i < a.V; i += 256) sm += wn_exp(srow[i] - mx);
  sm = wave_sum(sm);
  if (lane == 0) red[4 + wave] = sm;
  __syncthreads();
  const float lsum = logf(red[4] + red[5] + red[6] + red[7]);
  if (a.logp) {
    float* o = a.logp + (int64_t)row * a.ld_out;
    for (int i = tid; i < a.V; i += 256) o[i] = (srow[i] - mx) - lsum;
  }
  // top-k: every thread keeps the best of ITS strided elements; a round is one
  // block arg-max over those 256 candidates, after which only the winner's
  // owner rescans its V/256 elements (taken ones are overwritten with NaN,
  // which no comparison selects).  Order: larger value first, lower index on
  // ties (strict > while scanning indices upwards).
  __syncthreads();  // all reads of srow for the log-prob row are done
  auto local_best = [&]() {
    VI best;
    best.v = -INFINITY;
    best.i = 0x7fffffff;
    for (int i = tid; i < a.V; i += 256) {
      const float v = srow[i];
      if (v > best.v || (v == best.v && i < best.i)) { best.v = v; best.i = i; }
    }
    return best;
  };
  VI mine = local_best();
  for (int r = 0; r < a.k; ++r) {
    VI best = vi_wave(mine);
    if (lane == 0) redvi[wave] = best;
    __syncthreads();
    const VI b = vi_better(vi_better(redvi[0], redvi[1]),
                           vi_better(redvi[2], redvi[3]));
    if (tid == 0) {
      a.topk_val[(int64_t)row * a.k + r] = (b.v - mx) - lsum;
      a.topk_idx[(int64_t)row * a.k + r] = b.i;
    }
    if (b.i < a.V && (b.i & 255) == tid) {  // owner: retire it, find the next
      srow[b.i] = __builtin_nanf("");
      mine = local_best();
    }
    __syncthreads();  // redvi reusable
  }
}

// One WAVE per row, the row in registers (lane owns elements lane + 64 e): used when only
// the top-k is wanted (the decode path).  The block kernel above pays k block-wide
// arg-max rounds (two __syncthreads each) per row and ran at 0.9 TB/s.  Same order: larger
// value first, lower index on ties; a taken element becomes NaN, which no comparison selects.
// The sum runs in the order of the block kernel (its thread 64 w + lane owns the elements
// e = w mod 4 of this lane; four wave sums added left to right), so both kernels return
// bit-identical log-probs -- tests compare the decode path's top-k with searches on the full
// log-prob tensor bit for bit.
// Two levels (round 4 default; the one-level form -- a round scanned all EPL elements: k rounds
// x (EPL compares + selects, EPL retire compares) = ~3 k VALU instructions per row at EPL = 72,
// k = 10, 99 us per decode against 27 us for reading the logits once -- was bit-identical and is
// removed): the lane keeps the maximum (value, index) of each group of 8 of its elements; a
// round scans the EPL / 8 group maxima, and only the group the winner came from is retired and
// rescanned -- the winner is wave-uniform, so that is one scalar branch per group.  Same order
// everywhere (strict > while scanning indices upwards).  The selection order is restated lane
// by lane on the CPU against a stable top-k (tests/test_ctc_topk_form.py).
template <int EPL>
__global__ __launch_bounds__(256, EPL <= 72 ? 3 : 2) void ctc_row_wave2_kernel(CtcRowArgs a) {
  constexpr int G = 8, NG = EPL / G;
  static_assert(EPL % G == 0, "whole groups");
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.M) return;
  const float* x = a.logits + (int64_t)row * a.ld;
  float v[EPL];
  float mx = -INFINITY;
  // (clamped address + select instead of a guarded load: the guard compiles to a branch per
  // element, ~10 instructions each in the wave kernel's load phase)
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    const int i = e * 64 + lane;
    const float ld = x[min(i, a.V - 1)];
    float t = i < a.V ? ld : __builtin_nanf("");
    if (i == a.blank) t -= a.blank_penalty;
    v[e] = t;
    mx = fmaxf(mx, t);
  }
  mx = wave_max(mx);
  float sm4[4] = {0.f, 0.f, 0.f, 0.f};      // (the wave kernel's summation order)
#pragma unroll
  for (int e = 0; e < EPL; ++e)
    if (e * 64 + lane < a.V) sm4[e & 3] += wn_exp(v[e] - mx);
  const float lsum = logf(wave_sum(sm4[0]) + wave_sum(sm4[1]) + wave_sum(sm4[2]) +
                          wave_sum(sm4[3]));
  float gv[NG];
  int gi[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < G; ++j)
      if (v[g * G + j] > bv) { bv = v[g * G + j]; bi = (g * G + j) * 64 + lane; }
    gv[g] = bv;
    gi[g] = bi;
  }
  for (int r = 0; r < a.k; ++r) {
    VI best;
    best.v = -INFINITY;
    best.i = 0x7fffffff;
#pragma unroll
    for (int g = 0; g < NG; ++g)
      if (gv[g] > best.v) { best.v = gv[g]; best.i = gi[g]; }     // g ascending: lowest index
    const VI b = vi_wave(best);
    if (lane == 0) {
      a.topk_val[(int64_t)row * a.k + r] = (b.v - mx) - lsum;
      a.topk_idx[(int64_t)row * a.k + r] = b.i;
    }
    const int te = b.i >> 6;
    const bool mine = (b.i & 63) == lane;
    const int tg = __builtin_amdgcn_readfirstlane(te >> 3);   // every lane holds the same b
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (tg == g) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < G; ++j) {
          if (mine && g * G + j == te) v[g * G + j] = __builtin_nanf("");
          if (v[g * G + j] > bv) { bv = v[g * G + j]; bi = (g * G + j) * 64 + lane; }
        }
        gv[g] = bv;
        gi[g] = bi;
      }
    }
  }
}

// ===========================================================================
// ctc_greedy_search (search.py:109-124 + ctc_utils.py:23-33): frames past the
// utterance length count as blank, repeats collapse, blanks drop.  One wave per
// utterance, ballot-compaction 64 frames at a time.
__global__ __launch_bounds__(64) void greedy_kernel(
    const int* top1, int stride, const int* off, const int* len, int blank,
    int* out, int out_stride, int* out_lens) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int n = len[b], o = off[b];
  int count = 0;
  int carry = -1;  // token of frame t-1 (none before the first frame)
  for (int t0 = 0; t0 < n; t0 += 64) {
    const int t = t0 + lane;
    const int tok = t < n ? top1[(int64_t)(o + t) * stride] : blank;
    int prev = __shfl_up(tok, 1, 64);
    if (lane == 0) prev = carry;
    const bool keep = (t < n) && tok != blank && (t == 0 || tok != prev);
    const unsigned long long m = __ballot(keep);
    if (keep) {
      const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
      out[(int64_t)b * out_stride + pos] = tok;
    }
    count += __popcll(m);
    carry = __shfl(tok, 63, 64);
  }
  if (lane == 0) out_lens[b] = count;
}

// ===========================================================================
// CTC prefix beam search.
//
// The reference loops  for t: for u in topk(logp_t): for prefix in beam  and
// merges into a dict keyed by prefix.  Every key of that dict receives at most
// three contributions, all determined by the key itself:
//   blank  : key K, from hyp K                      (s, v_s, times_s)
//   repeat : key K, from hyp K, u = last(K)         (ns, v_ns, times_ns[-1]=t)
//   extend : key P+u, from hyp P                    (ns, v_ns, times + [t])
// and `extend` lands on an existing beam member K exactly when parent(K) = P
// and last(K) = u.  So one step is: evaluate <= beam + beam^2 independent
// entries in parallel (fp64, like Python floats), rank them by
// (score desc, dict-insertion order asc) -- Python's stable sort -- and keep
// `beam`.  The order-dependent corner of the reference (v_ns / cur_token_prob
// / times_ns when both `repeat` and `extend` hit one key) is replayed in the
// reference's own hyp order.  Prefixes and Viterbi time stamps are persistent
// linked lists in a per-utterance node pool, so no list is ever copied.
constexpr int MAXB = 16;
constexpr int MAXE = MAXB + MAXB * MAXB;
constexpr double NEG_INF = -HUGE_VAL;

// Prefix identity is the token sequence (the reference keys a dict by the
// tuple).  A physical node id is NOT an identity: a prefix that left the beam
// can be re-created later as a new node while its old children are still
// alive.  Every prefix therefore carries a 64-bit hash of its token sequence
// (chained splitmix64) -- as a FILTER only (round 3): two prefixes are the same iff their
// hashes are equal AND their token sequences are.  The sequence test is exact and almost
// always free: equal physical nodes are the same sequence (the common case: the parent of a
// beam member is usually still the very node it was created from); only hash-equal prefixes
// with DIFFERENT physical nodes -- a re-created prefix, or a 2^-64 collision -- are walked
// token by token through the node pool (same_prefix_nodes).  The pool is written with plain
// stores inside the frame loop; every wave drains its stores of frame t - 1 (s_waitcnt vmcnt(0),
// a frame after they were issued: no stall) before it writes frame t's and passes the frame's
// last barrier, and a walk only visits nodes created at least two frames ago, so what it
// reads (sc1 loads: L2, not a possibly stale L1 line) is complete.
typedef unsigned long long u64;
// weak (test knob "beam_weak_hash"): a 2-bit hash -- almost every pair of prefixes collides, so
// the results are right only if the exact sequence test behind the filter is
__device__ __forceinline__ u64 prefix_hash(u64 h, int tok, int weak = 0) {
  if (weak) return (h * 31u + (u64)tok) & 3u;
  u64 z = h ^ ((u64)(tok + 1) * 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
constexpr u64 ROOT_HASH = 0x243F6A8885A308D3ull;

// Token sequences of two prefix nodes equal?  Lock-step walk towards the root (node 0, parent
// -1): equal node ids end it (same node = same sequence), a different token or one side at the
// root first = different.
__device__ __noinline__ bool same_prefix_nodes(const int* n_parent, const int* n_token, int x,
                                               int y) {
  while (x != y) {
    if (x <= 0 || y <= 0) return false;
    const int tx = __hip_atomic_load(n_token + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int ty = __hip_atomic_load(n_token + y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tx != ty) return false;
    x = __hip_atomic_load(n_parent + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    y = __hip_atomic_load(n_parent + y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return true;
}

// The same walk with the node pool in LDS (prefix_beam_kernel<.., LPOOL = true>): plain reads,
// ordered by the frame loop's barriers; inlined so that the accesses stay ds_read.
__device__ __forceinline__ bool same_prefix_nodes_lds(const int* n_parent, const int* n_token,
                                                      int x, int y) {
  while (x != y) {
    if (x <= 0 || y <= 0) return false;
    if (n_token[x] != n_token[y]) return false;
    x = n_parent[x];
    y = n_parent[y];
  }
  return true;
}

// Latency design (the search is T' dependent steps per utterance; nothing
// here is bandwidth- or FLOP-bound):
//  * one workgroup per utterance with one THREAD per entry (beam + beam^2 <=
//    272): an entry lives in its thread's registers from evaluation to the
//    write of the new beam member; only (score, seq) go through LDS for the
//    rank;
//  * the beam is a structure of arrays in LDS, double buffered; every loop
//    over the beam is a broadcast LDS read;
//  * the stable rank is wave-parallel: lanes hold the entries j, a wave walks
//    its share of the entries e, and v_cmp + s_bcnt1 on the 64-bit compare mask
//    counts the entries that beat e -- 8 waves share the beam^2 x beam^2
//    comparison instead of one 110-iteration loop per thread;
//  * three barriers per frame, all LDS-only (s_waitcnt lgkmcnt(0) + s_barrier):
//    a __syncthreads() would also drain vmcnt, i.e. expose the HBM round trip
//    of the node-pool stores and of the top-k prefetch in every frame;
//  * the per-frame top-k (token, log-prob) pairs are staged through LDS 32
//    frames at a time, fetched one chunk ahead;
//  * the node pools are write-only inside the loop (depth and the
//    predecessor of the times_ns head travel with the beam member);
//  * log_add(a, b) = max + log(exp(a-max) + exp(b-max)): the max term is
//    exp(0) = 1 exactly and a -inf term adds exp(-inf) = 0 exactly, so the
//    fp64 exp / log run only for real merges -- bit-identical shortcuts.
struct HypSoA {  // the beam, structure of arrays
  u64 hash[MAXB], par_hash[MAXB];
  double s[MAXB], ns[MAXB], vs[MAXB], vns[MAXB];
  double score[MAXB], vit[MAXB];  // score() and viterbi_score()
  int node[MAXB], last[MAXB], par[MAXB], depth[MAXB];
  int ts[MAXB], tns[MAXB], tnsp[MAXB];  // times_s / times_ns heads (0 = empty
                                        // list), predecessor of the tns head
  int tim[MAXB];                        // times() head
  // context biasing (search.py:81-82): trie state and accumulated bonus
  double cscore[MAXB];
  int cstate[MAXB];
};

// ContextGraph.forward_one_step (context_graph.py:216-248): child of `state`
// by `token`, or along the fail arcs; returns the bonus, *next = new state.
__device__ __forceinline__ int ctx_child(const CtxGraph& g, int state, int token) {
  const u64 key = ((u64)(unsigned)state << 32) | (unsigned)token;
  unsigned h = ctx_slot(key, g.mask);
  while (true) {
    const u64 k = g.keys[h];
    if (k == key) return g.vals[h];
    if (k == CTX_EMPTY) return -1;
    h = (h + 1) & g.mask;
  }
}
__device__ __noinline__ double ctx_step(const CtxGraph& g, int state, int token, int* next) {
  int n = ctx_child(g, state, token);
  double score;
  if (n >= 0) {
    score = g.token_score[n];
  } else {
    n = g.fail[state];
    int c = ctx_child(g, n, token);
    while (c < 0) {
      n = g.fail[n];
      c = ctx_child(g, n, token);
      if (n == 0) break;
    }
    if (c >= 0) n = c;
    score = g.node_score[n] - g.node_score[state];
  }
  *next = n;
  return score + g.output_score[n];
}

// exp(d) for d in [-37, 0]: k = rint(d log2 e), r = d - k ln2 (two-part),
// degree-13 Taylor on |r| <= 0.347 (truncation 4e-18), scaled by 2^k.
__device__ __forceinline__ double exp_m37_0(double d) {
  const double k = rint(d * 1.4426950408889634);
  double r = fma(k, -6.93147180369123816490e-01, d);
  r = fma(k, -1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;           // 1/13!
  p = fma(p, r, 2.08767569878681e-09);         // 1/12!
  p = fma(p, r, 2.505210838544172e-08);        // 1/11!
  p = fma(p, r, 2.755731922398589e-07);        // 1/10!
  p = fma(p, r, 2.7557319223985893e-06);       // 1/9!
  p = fma(p, r, 2.48015873015873e-05);         // 1/8!
  p = fma(p, r, 1.984126984126984e-04);        // 1/7!
  p = fma(p, r, 1.388888888888889e-03);        // 1/6!
  p = fma(p, r, 8.333333333333333e-03);        // 1/5!
  p = fma(p, r, 4.1666666666666664e-02);       // 1/4!
  p = fma(p, r, 1.6666666666666666e-01);       // 1/3!
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)k);
}

// log(s) for s in [1, 2]: s = 2^e * m, m in [0.7071, 1.4142], f = m - 1
// (exact), z = f / (2 + f), log m = 2 atanh z = 2z (1 + w/3 + ... + w^9/19),
// w = z^2 <= 0.0295 (truncation 2e-17).
__device__ __forceinline__ double log_1_2(double s) {
  const bool hi = s > 1.4142135623730951;
  const double m = hi ? s * 0.5 : s;
  const double f = m - 1.0;
  const double z = f / (2.0 + f);
  const double w = z * z;
  double q = 5.2631578947368418e-02;           // 1/19
  q = fma(q, w, 5.8823529411764705e-02);       // 1/17
  q = fma(q, w, 6.6666666666666666e-02);       // 1/15
  q = fma(q, w, 7.6923076923076927e-02);       // 1/13
  q = fma(q, w, 9.0909090909090912e-02);       // 1/11
  q = fma(q, w, 1.1111111111111110e-01);       // 1/9
  q = fma(q, w, 1.4285714285714285e-01);       // 1/7
  q = fma(q, w, 2.0000000000000001e-01);       // 1/5
  q = fma(q, w, 3.3333333333333331e-01);       // 1/3
  const double z2 = z + z;
  const double r = fma(z2 * w, q, z2);
  return hi ? r + 6.9314718055994529e-01 : r;
}

// log_add (wenet/utils/common.py:302-310) of two values:
//   a_max + log(exp(a - a_max) + exp(b - a_max)).
// The a_max term is exp(0) = 1 exactly, a -inf term adds exp(-inf) = 0 exactly
// and below d = -37 the sum rounds to 1.0 (log -> 0), so those cases return
// a_max bit-identically to the reference; otherwise 1 + exp(d) and its log are
// evaluated with the two short fp64 kernels above (<= 2 ulp each; measured
// against math.log / math.exp in tests/test_gpu_ops.py) instead of the generic
// library exp / log: the search is bound by the length of this dependent
// chain, not by throughput.
__device__ __forceinline__ double log_add2_fast(double a, double b) {
  const double m = a > b ? a : b;
  const double n = a > b ? b : a;
  const double d = n - m;          // <= 0, or NaN for (-inf, -inf)
  if (!(d >= -37.0)) return m;     // covers n == -inf and (-inf, -inf) -> -inf
  return m + log_1_2(1.0 + exp_m37_0(d));
}

__global__ void log_add_kernel(const double* a, const double* b, double* o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = log_add2_fast(a[i], b[i]);
}

// Workgroup barrier that orders LDS traffic only (global loads / stores stay
// in flight across it).
__device__ __forceinline__ void barrier_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

constexpr int PB_CHUNK = 32;  // frames of top-k staged per LDS buffer
constexpr int PB_THREADS = 512;
constexpr int PB_WAVES = PB_THREADS / 64;
constexpr int PB_CHUNKS = (MAXE + 63) / 64;  // 5: beam up to 16
constexpr size_t PB_LPOOL_BYTES = 128 * 1024;   // dynamic LDS the node pool may take (static: ~16 KB)

// NCH = 64-entry chunks the rank pass scans: compile-time so that the pass has
// no per-chunk branches (2 covers beam <= 10, the default of every recipe).
// CTX: context biasing compiled in (a.cg set); the plain search keeps its own
// instantiation so that its frame loop carries none of this.
// LPOOL: the utterance's node pool (4 x (max_len * beam + 1) ints) lives in dynamic LDS instead
// of the global scratch: the pool stores of the frame loop, the rare sequence walks and above all
// the n-best emission (one DEPENDENT load per token and hypothesis: 113 us of the 1.17 ms search
// at config 2 through L2, r06x) run at LDS latency.  The launcher takes it whenever the pool fits
// (PB_LPOOL_BYTES), i.e. up to ~32 s of audio at beam 10.
template <int NCH, bool CTX, bool LPOOL>
__global__ __launch_bounds__(PB_THREADS) void prefix_beam_kernel(PrefixBeamArgs a) {
  extern __shared__ __attribute__((aligned(16))) int lds_pool[];
  // NCH = 2 <=> MAXB + beam^2 <= 128 <=> beam <= 10: the scans over the beam stop there
  constexpr int BMAX = NCH == 2 ? 10 : MAXB;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63;
  // wave-uniform values must be PROVABLY uniform (SGPRs) or every loop on
  // them is compiled as a divergent, exec-masked loop
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = a.len[b], off = a.off[b];
  const int beam = a.beam;
  __shared__ HypSoA hyp[2];
  __shared__ double e_score[MAXE];
  __shared__ int e_seq[MAXE];
  __shared__ int e_rank[MAXE];
  __shared__ int s_nvalid[2];
  __shared__ int s_claim[2][MAXB];
  __shared__ int s_tie[2];
  __shared__ int tok[2][PB_CHUNK][MAXB];
  __shared__ float lp[2][PB_CHUNK][MAXB];

  const int cap = a.max_len * beam + 1;
  int* pool = LPOOL ? lds_pool : a.pool + (int64_t)b * a.pool_stride;
  int* n_parent = pool;            // prefix nodes
  int* n_token = pool + cap;
  int* t_prev = pool + 2 * cap;    // time nodes
  int* t_val = pool + 3 * cap;
  auto same_nodes = [&](int x, int y) __attribute__((always_inline)) {
    if constexpr (LPOOL) return same_prefix_nodes_lds(n_parent, n_token, x, y);
    else return same_prefix_nodes(n_parent, n_token, x, y);
  };

  if (tid == 0) {
    n_parent[0] = -1; n_token[0] = -1;
    t_prev[0] = 0; t_val[0] = -1;
    HypSoA& h = hyp[0];
    h.hash[0] = ROOT_HASH; h.par_hash[0] = 0;
    h.node[0] = 0; h.last[0] = -1; h.par[0] = -1; h.depth[0] = 0;
    h.ts[0] = 0; h.tns[0] = 0; h.tnsp[0] = 0; h.tim[0] = 0;
    h.s[0] = 0.0; h.ns[0] = NEG_INF; h.vs[0] = 0.0; h.vns[0] = 0.0;  // search.py:144-147
    h.score[0] = 0.0; h.vit[0] = 0.0;
    h.cstate[0] = 0; h.cscore[0] = 0.0;   // search.py:148-150: root, no bonus
    s_nvalid[0] = 0; s_nvalid[1] = 0;
    s_tie[0] = 0; s_tie[1] = 0;
  }
  if (tid < 2 * MAXB) s_claim[tid / MAXB][tid % MAXB] = 0;
  {  // top-k chunk 0
    const int f = tid / beam, q = tid - f * beam;
    if (f < PB_CHUNK && f < T) {
      tok[0][f][q] = a.topk_idx[(int64_t)(off + f) * a.k + q];
      lp[0][f][q] = a.topk_val[(int64_t)(off + f) * a.k + q];
    }
  }
  __syncthreads();

  // Entry slots are fixed for the whole search so that no index arithmetic
  // (integer division by `beam`) sits in the frame loop, and the two kinds of
  // entries never share a wave (no divergent double pass):
  //   slot r           (wave 0, lanes < BMAX) : unchanged prefix H[r]
  //   slot BMAX + r*beam + q  (waves 1..)     : extension H[r] + topk[q]
  // (BMAX = 10 for beam <= 10: 110 slots, 14 per wave in the rank pass)
  const int x_r = tid >= 64 ? (tid - 64) / beam : 0;
  const int x_q = tid >= 64 ? (tid - 64) - x_r * beam : 0;
  const int my_slot = tid < BMAX ? tid : (tid >= 64 ? BMAX + (tid - 64) : 0x7fffffff);
  const int nx_f = tid / beam, nx_q = tid - nx_f * beam;

  const bool dbg = a.dbg_cycles != nullptr && b == 0 && tid == 0;
  long long c_eval = 0, c_rank = 0, c_sel = 0;
  int cur = 0, nb = 1;
  for (int t = 0; t < T; ++t) {
    const long long c0 = dbg ? __builtin_amdgcn_s_memtime() : 0;
    const HypSoA& H = hyp[cur];
    const int* tk = tok[(t / PB_CHUNK) & 1][t % PB_CHUNK];
    const float* lq = lp[(t / PB_CHUNK) & 1][t % PB_CHUNK];
    // first frame of a chunk: fetch the NEXT chunk now, park it in LDS at the
    // end of this frame (its buffer was last read in the previous frame)
    int nx_tok = 0;
    float nx_lp = 0.f;
    const int nx_t = (t / PB_CHUNK + 1) * PB_CHUNK + nx_f;
    const bool nx_on = (t % PB_CHUNK == 0) && nx_f < PB_CHUNK && nx_t < T;
    if (nx_on) {
      nx_tok = a.topk_idx[(int64_t)(off + nx_t) * a.k + nx_q];
      nx_lp = a.topk_val[(int64_t)(off + nx_t) * a.k + nx_q];
    }
    const int n_ent = BMAX + nb * beam;  // slots in use (with holes)
    // ---- this thread's entry ----------------------------------------------
    int valid = 0;
    double Es = NEG_INF, Ens = NEG_INF, Evs = NEG_INF, Evns = NEG_INF;
    int Ets = 0, Etns_src = 0, Etns_op = 0, Etnsp = 0;  // op: 0 empty, 1 append t, 2 replace last
    int Ekey = -1, Epar = -1, Etoken = -1, Edepth = 0, seq = 0x7fffffff;
    u64 Ehash = 0, Eparh = 0;
    double Ecx = 0.0;  // context bonus / state of the entry: set by the FIRST
    int Ecs = 0;       // contribution in the reference's loop order (has_context)
    if (tid < nb) {
      // ---- unchanged prefix K = H[r] ------------------------------------------
      // Every LDS read whose address is known up front is issued here, unconditionally (slots
      // >= beam / nb hold stale values and are masked in the compares): ONE round trip for the
      // scans and for K's own fields instead of one per use; the only dependent reads left are
      // the parent's fields (index rp), again one batch.
      const int r = tid;
      const int Klast = H.last[r];
      Ehash = H.hash[r]; Eparh = H.par_hash[r];
      const int Knode = H.node[r], Kpar = H.par[r], Kdepth = H.depth[r];
      const double Kscore = H.score[r], Kvit = H.vit[r], Kns = H.ns[r], Kvns = H.vns[r];
      const int Ktim = H.tim[r], Ktns = H.tns[r], Ktnsp = H.tnsp[r];
      int tq[BMAX], nd[BMAX];
      float lv[BMAX];
      u64 hh[BMAX];
#pragma unroll
      for (int q = 0; q < BMAX; ++q) { tq[q] = tk[q]; lv[q] = lq[q]; }
#pragma unroll
      for (int j = 0; j < BMAX; ++j) { hh[j] = H.hash[j]; nd[j] = H.node[j]; }
      int qb = -1, ql = -1;
      float pb = 0.f, pl = 0.f;
#pragma unroll
      for (int q = 0; q < BMAX; ++q) {
        const bool inb = q < beam;
        const bool mb = inb & (tq[q] == a.blank);
        const bool ml = inb & (Klast >= 0) & (tq[q] == Klast);
        qb = mb ? q : qb; pb = mb ? lv[q] : pb;
        ql = ml ? q : ql; pl = ml ? lv[q] : pl;
      }
      // parent of K inside the beam: hash filter, then the exact test -- the same node (the
      // usual case, decided in the scan), else the walk
      unsigned rp_hits = 0, rp_same = 0;
#pragma unroll
      for (int j = 0; j < BMAX; ++j) {
        const bool hm = (j < nb) & (hh[j] == Eparh);
        rp_hits |= hm ? (1u << j) : 0u;
        rp_same |= (hm & (nd[j] == Kpar)) ? (1u << j) : 0u;
      }
      unsigned rp_set = rp_same;
      for (unsigned mk = rp_hits & ~rp_same; mk; mk &= mk - 1) {   // rare: a re-created prefix
        const int j = __ffs(mk) - 1;
        if (same_nodes(H.node[j], Kpar)) rp_set |= 1u << j;
      }
      const int rp = rp_set ? 31 - __clz(rp_set) : -1;   // (the last match in beam order)
      // the parent's fields, one batch (slot 0 when there is none: never used then)
      const int rpc = max(rp, 0);
      const int Plast = H.last[rpc], Pts = H.ts[rpc], Ptim = H.tim[rpc];
      const double Ps = H.s[rpc], Pvs = H.vs[rpc], Pscore = H.score[rpc], Pvit = H.vit[rpc];
      if (qb >= 0 || ql >= 0) {
        valid = 1;
        Ekey = Knode; Epar = Kpar; Etoken = Klast; Edepth = Kdepth;
        if (qb >= 0) {
          const double p = (double)pb;
          Es = Kscore + p;       // log_add(-inf, x) == x
          Evs = Kvit + p;
          Ets = Ktim;
          seq = min(seq, (qb * nb + r) * 2);
        }
        if (ql >= 0) {
          const double p = (double)pl;
          const int u = Klast;
          const double xa = Kns + p, va = Kvns + p;
          Etnsp = Ktnsp;
          seq = min(seq, (ql * nb + r) * 2);
          double v = NEG_INF, ctp = NEG_INF;
          int tsrc = 0, top = 0;
          if (rp < 0) {
            Ens = xa;
            if (v < va) { v = va; tsrc = Ktns; top = 2; }
          } else {
            const bool rep = Plast == u;
            const double xb = (rep ? Ps : Pscore) + p, vb = (rep ? Pvs : Pvit) + p;
            const int tb = rep ? Pts : Ptim, sub = rep ? 1 : 0;
            seq = min(seq, (ql * nb + rp) * 2 + sub);
            Ens = log_add2_fast(xa, xb);
            if (r < rp) {          // hyp K is visited before its parent
              if (v < va) { v = va; ctp = p; tsrc = Ktns; top = 2; }
              if (v < vb) { v = vb; ctp = p; tsrc = tb; top = 1; }
            } else {               // parent first
              if (v < vb) { v = vb; ctp = p; tsrc = tb; top = 1; }
              if (v < va) {
                v = va;
                if (ctp < p) { ctp = p; tsrc = Ktns; top = 2; }
              }
            }
          }
          Evns = v; Etns_src = tsrc; Etns_op = top;
        }
        if (CTX) {
          // visits of key K in loop order (token rank outer, hyp inner): K's own
          // blank (qb, r) and repeat (ql, r) copy K's context, the parent's
          // extension (ql, rp) walks the graph from the parent's state
          int first = 0x7fffffff;
          bool from_parent = false;
          if (qb >= 0) first = qb * nb + r;
          if (ql >= 0) {
            first = min(first, ql * nb + r);
            if (rp >= 0 && ql * nb + rp < first) from_parent = true;
          }
          if (from_parent) {
            Ecx = H.cscore[rp] + ctx_step(a.cg, H.cstate[rp], Klast, &Ecs);
          } else {
            Ecx = H.cscore[r]; Ecs = H.cstate[r];
          }
        }
      }
    } else if (tid >= 64 && x_r < nb) {
      // ---- extension P + u ----------------------------------------------------
      // the same rule: P's fields, the token, the beam's hashes / last tokens / parents in one
      // batch of unconditional reads
      const int r = x_r, q = x_q;
      const int u = tk[q];
      const float lpq = lq[q];
      const u64 Ph = H.hash[r];
      const int Pn = H.node[r], Plast = H.last[r], Pdepth = H.depth[r];
      const int Pts = H.ts[r], Ptim = H.tim[r];
      const double Ps = H.s[r], Pvs = H.vs[r], Pscore = H.score[r], Pvit = H.vit[r];
      u64 hh[BMAX];
      int lj[BMAX], pj[BMAX];
#pragma unroll
      for (int j = 0; j < BMAX; ++j) { hh[j] = H.hash[j]; lj[j] = H.last[j]; pj[j] = H.par[j]; }
      const u64 ch = prefix_hash(Ph, u, a.weak_hash);
      // does P + u land on a beam member?  hash filter, then exact: same last token and the
      // member's parent IS P (same node -- decided in the scan -- else the walk)
      unsigned mg_walk = 0;
      bool merged = false;
#pragma unroll
      for (int j = 0; j < BMAX; ++j) {
        const bool hm = (j < nb) & (hh[j] == ch) & (lj[j] == u);
        merged |= hm & (pj[j] == Pn);
        mg_walk |= (hm & (pj[j] != Pn)) ? (1u << j) : 0u;
      }
      if (!merged) {
        for (unsigned mk = mg_walk; mk; mk &= mk - 1) {     // rare: a re-created prefix
          const int j = __ffs(mk) - 1;
          merged |= same_nodes(H.par[j], Pn);
        }
      }
      if (u != a.blank && !merged) {
        const double p = (double)lpq;
        const bool rep = u == Plast;
        const double x = (rep ? Ps : Pscore) + p, v = (rep ? Pvs : Pvit) + p;
        const int tb = rep ? Pts : Ptim, sub = rep ? 1 : 0;
        valid = 1;
        Ens = x;
        if (v > NEG_INF) { Evns = v; Etns_src = tb; Etns_op = 1; }
        Ekey = -1; Epar = Pn; Etoken = u; Edepth = Pdepth + 1;
        Ehash = ch; Eparh = Ph;
        seq = (q * nb + r) * 2 + sub;
        if (CTX) Ecx = H.cscore[r] + ctx_step(a.cg, H.cstate[r], u, &Ecs);
      }
    }
    const double Escore = log_add2_fast(Es, Ens);
    // second prune key: total_score() = score() + context_score (search.py:93-94)
    const double Etotal = CTX ? Escore + Ecx : Escore;
    if (my_slot < n_ent) {
      e_score[my_slot] = valid ? Etotal : NEG_INF;
      e_seq[my_slot] = valid ? seq : 0x7fffffff;
    }
    {
      const unsigned long long vm = __ballot(valid);
      if (lane == 0 && vm) atomicAdd(&s_nvalid[t & 1], __popcll(vm));
    }
    barrier_lds();
    const long long c1 = dbg ? __builtin_amdgcn_s_memtime() : 0;
    const int n_valid = __builtin_amdgcn_readfirstlane(s_nvalid[t & 1]);
    // ---- second beam prune: stable sort by score desc, keep `beam` -----------
    // rank(e) = #{ j : score_j > score_e or (score_j == score_e and seq_j < seq_e) }
    // Fast pass: count only score_j > score_e (one v_cmp + s_bcnt1 per 64
    // entries).  That is the exact rank of every entry whose score is unique;
    // entries that tie get the same number, so a tie inside the kept range
    // shows up as two claims on one rank -> that frame reruns the exact pass.
    auto rank_pass = [&](const bool exact) {
      double sj[NCH];
      int qj[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int j = c * 64 + lane;
        const bool in = j < n_ent;
        sj[c] = in ? e_score[j] : NEG_INF;
        qj[c] = in ? e_seq[j] : 0x7fffffff;
      }
      // this wave ranks the slots e = wave + 8*it; lane `it` fetches slot e's
      // key once, the loop broadcasts it with v_readlane (no LDS round trip
      // per iteration) and lane `it` keeps the result
      const int e_l = wave + PB_WAVES * lane;
      const bool e_in = e_l < n_ent;
      const double se_v = e_in ? e_score[e_l] : NEG_INF;
      const int qe_v = e_in ? e_seq[e_l] : 0x7fffffff;
      const int se_lo = __double2loint(se_v), se_hi = __double2hiint(se_v);
      int my_r = 0;
      auto rank_of = [&](const double se, const int qe) __attribute__((always_inline)) {
        int r = 0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          if (exact) {
            // bitwise on purpose: no short-circuit branches in the loop
            const bool beats = (sj[c] > se) | ((sj[c] == se) & (qj[c] < qe));
            r += __popcll(__ballot(beats));
          } else {
            r += __popcll(__ballot(sj[c] > se));
          }
        }
        return r;
      };
      if constexpr (NCH == 2) {
        // <= 110 slots = 14 per wave: fully unrolled with constant lane indices (v_readlane with
        // an immediate, v_writelane of the count into lane `it`): 16 independent chains instead
        // of a rolled loop whose every step waits on VALU -> SGPR -> SALU -> VALU hazards
        // (2550 -> ~1500 cycles of the frame's 8400, r06z).  Slots past n_ent carry -inf keys:
        // their counts are computed and never stored.
#pragma unroll
        for (int it = 0; it < (BMAX + BMAX * BMAX + PB_WAVES - 1) / PB_WAVES; ++it) {
          const double se = __hiloint2double(__builtin_amdgcn_readlane(se_hi, it),
                                             __builtin_amdgcn_readlane(se_lo, it));
          const int qe = __builtin_amdgcn_readlane(qe_v, it);
          const int r = __builtin_amdgcn_readfirstlane(rank_of(se, qe));   // (uniform: an SGPR)
          asm("v_writelane_b32 %0, %1, %2" : "+v"(my_r) : "s"(r), "n"(it));
        }
      } else {
        const int n_it = (n_ent - wave + PB_WAVES - 1) / PB_WAVES;
#pragma unroll 4
        for (int it = 0; it < n_it; ++it) {
          const double se = __hiloint2double(__builtin_amdgcn_readlane(se_hi, it),
                                             __builtin_amdgcn_readlane(se_lo, it));
          const int qe = __builtin_amdgcn_readlane(qe_v, it);
          const int r = rank_of(se, qe);
          if (lane == it) my_r = r;
        }
      }
      if (e_in) {
        e_rank[e_l] = my_r;
        if (!exact && qe_v != 0x7fffffff && my_r < beam) {
          if (atomicAdd(&s_claim[t & 1][my_r], 1) != 0) s_tie[t & 1] = 1;
        }
      }
    };
    rank_pass(false);
    barrier_lds();
    if (__builtin_amdgcn_readfirstlane(s_tie[t & 1])) {
      rank_pass(true);
      barrier_lds();
    }
    const long long c2 = dbg ? __builtin_amdgcn_s_memtime() : 0;
    if (tid < MAXB) s_claim[(t + 1) & 1][tid] = 0;
    if (tid == 0) { s_nvalid[(t + 1) & 1] = 0; s_tie[(t + 1) & 1] = 0; }
    // the node-pool stores of the PREVIOUS frame are complete before this frame's are issued
    // (they had a whole frame: no stall) -- see the prefix-identity note above
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (valid) {
      const int rank = e_rank[my_slot];
      if (rank < beam) {
        HypSoA& Hn = hyp[cur ^ 1];
        const int slot = 1 + t * beam + rank;
        int node = Ekey;
        if (Ekey < 0) {
          node = slot;
          n_parent[slot] = Epar;
          n_token[slot] = Etoken;
        }
        int tns = 0, tnsp = 0;
        if (Etns_op == 1) {
          t_prev[slot] = Etns_src; t_val[slot] = t; tns = slot; tnsp = Etns_src;
        } else if (Etns_op == 2) {
          t_prev[slot] = Etnsp; t_val[slot] = t; tns = slot; tnsp = Etnsp;
        }
        Hn.node[rank] = node; Hn.par[rank] = Epar; Hn.last[rank] = Etoken;
        Hn.depth[rank] = Edepth;
        Hn.hash[rank] = Ehash; Hn.par_hash[rank] = Eparh;
        Hn.s[rank] = Es; Hn.ns[rank] = Ens; Hn.vs[rank] = Evs; Hn.vns[rank] = Evns;
        Hn.ts[rank] = Ets; Hn.tns[rank] = tns; Hn.tnsp[rank] = tnsp;
        Hn.score[rank] = Escore;
        Hn.vit[rank] = Evs > Evns ? Evs : Evns;   // search.py:87-88
        Hn.tim[rank] = Evs > Evns ? Ets : tns;    // search.py:90-91
        if (CTX) { Hn.cscore[rank] = Ecx; Hn.cstate[rank] = Ecs; }
      }
    }
    if (nx_on) {
      tok[(t / PB_CHUNK + 1) & 1][nx_f][nx_q] = nx_tok;
      lp[(t / PB_CHUNK + 1) & 1][nx_f][nx_q] = nx_lp;
    }
    nb = min(beam, n_valid);
    cur ^= 1;
    barrier_lds();
    if (dbg) {
      const long long c3 = __builtin_amdgcn_s_memtime();
      c_eval += c1 - c0; c_rank += c2 - c1; c_sel += c3 - c2;
    }
  }
  const long long c_loop_end = dbg ? __builtin_amdgcn_s_memtime() : 0;
  __syncthreads();  // node-pool stores visible to the emitting threads

  // ---- emit the n-best list -------------------------------------------------
  if (tid == 0) a.n_hyps[b] = nb;
  if (tid < beam) {
    const int64_t o = (int64_t)b * beam + tid;
    if (tid < nb) {
      const HypSoA& h = hyp[cur];
      const int L = h.depth[tid];
      a.hyp_lens[o] = L;
      // search.py:229-237: finalize() REPLACES the bonus by -node_score(state)
      a.hyp_scores[o] = CTX ? h.score[tid] + (-a.cg.node_score[h.cstate[tid]])
                            : h.score[tid];
      // The token list and the time list are walked in ONE loop (two
      // independent chains of dependent loads overlap).  times() is either
      // never set (head 0) or has one entry per token; anything else falls
      // back to the two-pass walk.
      int* tkn = a.hyp_tokens + o * a.max_len;
      int* tm = a.hyp_times + o * a.max_len;
      int node = h.node[tid];
      int x = h.tim[tid];
      const bool has_t = x != 0;
      int cnt = 0;
      for (int i = L - 1; i >= 0; --i) {
        tkn[i] = n_token[node];
        node = n_parent[node];
        if (x != 0) { tm[i] = t_val[x]; x = t_prev[x]; ++cnt; }
      }
      int n_t = has_t ? L : 0;
      if (has_t && (x != 0 || cnt != L)) {  // not one time per token: recount
        n_t = 0;
        for (int y = h.tim[tid]; y != 0; y = t_prev[y]) ++n_t;
        int y = h.tim[tid];
        for (int i = n_t - 1; i >= 0; --i) { tm[i] = t_val[y]; y = t_prev[y]; }
      }
      a.hyp_tlens[o] = n_t;  // == L, or 0 for a never-set list
    } else {
      a.hyp_lens[o] = 0;
      a.hyp_tlens[o] = 0;
      a.hyp_scores[o] = NEG_INF;
    }
  }
  if (dbg) {
    a.dbg_cycles[0] = c_eval; a.dbg_cycles[1] = c_rank; a.dbg_cycles[2] = c_sel;
    a.dbg_cycles[3] = T;
    a.dbg_cycles[4] = __builtin_amdgcn_s_memtime() - c_loop_end;
  }
}

// ===========================================================================
// Large beams (17..64): the same per-entry formulation, written for generality
// instead of latency.  One 1024-thread workgroup per utterance; the
// <= 64 + 64*64 entries of a frame are dealt to the threads (<= 5 each), every
// thread keeps only (total score, seq) of its entries, the `beam` survivors are
// picked by `beam` rounds of a block-wide arg-max on (score desc, seq asc) --
// exactly Python's stable sort order -- and only the survivors are evaluated in
// full and written.  The default recipes (beam 10) never come here.
constexpr int BIGB = 64;
constexpr int BIG_THREADS = 1024;
constexpr int BIG_WAVES = BIG_THREADS / 64;
constexpr int BIG_EPT = (BIGB + BIGB * BIGB + BIG_THREADS - 1) / BIG_THREADS;

struct BigHyp {
  u64 hash[BIGB], par_hash[BIGB];
  double s[BIGB], ns[BIGB], vs[BIGB], vns[BIGB], score[BIGB], vit[BIGB], cscore[BIGB];
  int node[BIGB], last[BIGB], par[BIGB], depth[BIGB];
  int ts[BIGB], tns[BIGB], tnsp[BIGB], tim[BIGB], cstate[BIGB];
};

struct BigEntry {
  int valid;
  double s, ns, vs, vns, cx;
  int ts, tns_src, tns_op, tnsp;
  int key, par, token, depth, seq, cs;
  u64 hash, parh;
};

// Entry e of the frame: e < BIGB is the unchanged prefix H[e]; otherwise the
// extension H[r] + topk[q] with (r, q) = divmod(e - BIGB, beam).  Same
// arithmetic, same order rules as prefix_beam_kernel above.
template <bool CTX>
__device__ BigEntry big_eval(const BigHyp& H, int nb, int beam, int blank,
                             const int* tk, const float* lq, int e, const CtxGraph& cg,
                             const int* n_parent, const int* n_token, int weak_hash) {
  BigEntry E;
  E.valid = 0;
  E.s = E.ns = E.vs = E.vns = NEG_INF; E.cx = 0.0;
  E.ts = E.tns_src = E.tns_op = E.tnsp = 0;
  E.key = E.par = E.token = -1; E.depth = 0; E.seq = 0x7fffffff; E.cs = 0;
  E.hash = E.parh = 0;
  if (e < BIGB) {
    const int r = e;
    if (r >= nb) return E;
    const int Klast = H.last[r];
    E.hash = H.hash[r]; E.parh = H.par_hash[r];
    int qb = -1, ql = -1, rp = -1;
    for (int q = 0; q < beam; ++q) {
      if (tk[q] == blank) qb = q;
      if (Klast >= 0 && tk[q] == Klast) ql = q;
    }
    for (int j = 0; j < nb; ++j)
      if (H.hash[j] == E.parh &&
          (H.node[j] == H.par[r] || same_prefix_nodes(n_parent, n_token, H.node[j], H.par[r])))
        rp = j;
    if (qb < 0 && ql < 0) return E;
    E.valid = 1;
    E.key = H.node[r]; E.par = H.par[r]; E.token = Klast; E.depth = H.depth[r];
    if (qb >= 0) {
      const double p = (double)lq[qb];
      E.s = H.score[r] + p;
      E.vs = H.vit[r] + p;
      E.ts = H.tim[r];
      E.seq = min(E.seq, (qb * nb + r) * 2);
    }
    if (ql >= 0) {
      const double p = (double)lq[ql];
      const int u = Klast;
      const double xa = H.ns[r] + p, va = H.vns[r] + p;
      const int Ktns = H.tns[r];
      E.tnsp = H.tnsp[r];
      E.seq = min(E.seq, (ql * nb + r) * 2);
      double v = NEG_INF, ctp = NEG_INF;
      int tsrc = 0, top = 0;
      if (rp < 0) {
        E.ns = xa;
        if (v < va) { v = va; tsrc = Ktns; top = 2; }
      } else {
        double xb, vb; int tb, sub;
        if (H.last[rp] == u) { xb = H.s[rp] + p; vb = H.vs[rp] + p; tb = H.ts[rp]; sub = 1; }
        else { xb = H.score[rp] + p; vb = H.vit[rp] + p; tb = H.tim[rp]; sub = 0; }
        E.seq = min(E.seq, (ql * nb + rp) * 2 + sub);
        E.ns = log_add2_fast(xa, xb);
        if (r < rp) {
          if (v < va) { v = va; ctp = p; tsrc = Ktns; top = 2; }
          if (v < vb) { v = vb; ctp = p; tsrc = tb; top = 1; }
        } else {
          if (v < vb) { v = vb; ctp = p; tsrc = tb; top = 1; }
          if (v < va) {
            v = va;
            if (ctp < p) { ctp = p; tsrc = Ktns; top = 2; }
          }
        }
      }
      E.vns = v; E.tns_src = tsrc; E.tns_op = top;
    }
    if (CTX) {
      int first = 0x7fffffff;
      bool from_parent = false;
      if (qb >= 0) first = qb * nb + r;
      if (ql >= 0) {
        first = min(first, ql * nb + r);
        if (rp >= 0 && ql * nb + rp < first) from_parent = true;
      }
      if (from_parent) E.cx = H.cscore[rp] + ctx_step(cg, H.cstate[rp], Klast, &E.cs);
      else { E.cx = H.cscore[r]; E.cs = H.cstate[r]; }
    }
    return E;
  }
  const int x = e - BIGB;
  const int r = x / beam, q = x - r * beam;
  if (r >= nb) return E;
  const int u = tk[q];
  if (u == blank) return E;
  const u64 Ph = H.hash[r];
  const u64 ch = prefix_hash(Ph, u, weak_hash);
  for (int j = 0; j < nb; ++j)
    if (H.hash[j] == ch && H.last[j] == u &&
        (H.par[j] == H.node[r] || same_prefix_nodes(n_parent, n_token, H.par[j], H.node[r])))
      return E;  // lands on a beam member: that entry owns it
  const double p = (double)lq[q];
  double xx, v; int tb, sub;
  if (u == H.last[r]) { xx = H.s[r] + p; v = H.vs[r] + p; tb = H.ts[r]; sub = 1; }
  else { xx = H.score[r] + p; v = H.vit[r] + p; tb = H.tim[r]; sub = 0; }
  E.valid = 1;
  E.ns = xx;
  if (v > NEG_INF) { E.vns = v; E.tns_src = tb; E.tns_op = 1; }
  E.key = -1; E.par = H.node[r]; E.token = u; E.depth = H.depth[r] + 1;
  E.hash = ch; E.parh = Ph;
  E.seq = (q * nb + r) * 2 + sub;
  if (CTX) E.cx = H.cscore[r] + ctx_step(cg, H.cstate[r], u, &E.cs);
  return E;
}

struct BigKey { double s; int q; int e; };
__device__ __forceinline__ bool big_better(const BigKey& a, const BigKey& b) {
  // does b beat a?  larger score first, then the earlier dict insertion
  return (b.s > a.s) || (b.s == a.s && b.q < a.q);
}

template <bool CTX>
__global__ __launch_bounds__(BIG_THREADS) void prefix_beam_big_kernel(PrefixBeamArgs a) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = a.len[b], off = a.off[b];
  const int beam = a.beam;
  __shared__ BigHyp hyp[2];
  __shared__ int tk[BIGB];
  __shared__ float lq[BIGB];
  __shared__ double w_s[2][BIG_WAVES];
  __shared__ int w_q[2][BIG_WAVES], w_e[2][BIG_WAVES];

  const int cap = a.max_len * beam + 1;
  int* pool = a.pool + (int64_t)b * a.pool_stride;
  int* n_parent = pool;
  int* n_token = pool + cap;
  int* t_prev = pool + 2 * cap;
  int* t_val = pool + 3 * cap;
  if (tid == 0) {
    n_parent[0] = -1; n_token[0] = -1;
    t_prev[0] = 0; t_val[0] = -1;
    BigHyp& h = hyp[0];
    h.hash[0] = ROOT_HASH; h.par_hash[0] = 0;
    h.node[0] = 0; h.last[0] = -1; h.par[0] = -1; h.depth[0] = 0;
    h.ts[0] = 0; h.tns[0] = 0; h.tnsp[0] = 0; h.tim[0] = 0;
    h.s[0] = 0.0; h.ns[0] = NEG_INF; h.vs[0] = 0.0; h.vns[0] = 0.0;
    h.score[0] = 0.0; h.vit[0] = 0.0;
    h.cstate[0] = 0; h.cscore[0] = 0.0;
  }
  int cur = 0, nb = 1;
  for (int t = 0; t < T; ++t) {
    if (tid < beam) {
      tk[tid] = a.topk_idx[(int64_t)(off + t) * a.k + tid];
      lq[tid] = a.topk_val[(int64_t)(off + t) * a.k + tid];
    }
    __syncthreads();  // top-k of this frame, beam written by the previous frame
    const BigHyp& H = hyp[cur];
    const int n_ent = BIGB + nb * beam;
    double ks[BIG_EPT];
    int kq[BIG_EPT], kr[BIG_EPT];
#pragma unroll
    for (int i = 0; i < BIG_EPT; ++i) {
      const int e = tid + i * BIG_THREADS;
      ks[i] = NEG_INF; kq[i] = 0x7fffffff; kr[i] = -1;
      if (e < n_ent) {
        const BigEntry E = big_eval<CTX>(H, nb, beam, a.blank, tk, lq, e, a.cg, n_parent, n_token, a.weak_hash);
        if (E.valid) {
          const double sc = log_add2_fast(E.s, E.ns);
          ks[i] = CTX ? sc + E.cx : sc;   // total_score(), search.py:93-94
          kq[i] = E.seq;
        }
      }
    }
    // `beam` rounds of block arg-max; every thread derives the same winner
    int n_sel = 0;
    for (int k = 0; k < beam; ++k) {
      BigKey best; best.s = NEG_INF; best.q = 0x7fffffff; best.e = -1;
#pragma unroll
      for (int i = 0; i < BIG_EPT; ++i) {
        BigKey c; c.s = ks[i]; c.q = kq[i]; c.e = tid + i * BIG_THREADS;
        if (kq[i] != 0x7fffffff && (best.e < 0 || big_better(best, c))) best = c;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        BigKey c;
        c.s = __shfl_xor(best.s, o, 64);
        c.q = __shfl_xor(best.q, o, 64);
        c.e = __shfl_xor(best.e, o, 64);
        if (c.e >= 0 && (best.e < 0 || big_better(best, c))) best = c;
      }
      if (lane == 0) { w_s[k & 1][wave] = best.s; w_q[k & 1][wave] = best.q; w_e[k & 1][wave] = best.e; }
      __syncthreads();
      BigKey win; win.s = NEG_INF; win.q = 0x7fffffff; win.e = -1;
      for (int w = 0; w < BIG_WAVES; ++w) {
        BigKey c; c.s = w_s[k & 1][w]; c.q = w_q[k & 1][w]; c.e = w_e[k & 1][w];
        if (c.e >= 0 && (win.e < 0 || big_better(win, c))) win = c;
      }
      if (win.e < 0) break;  // fewer valid entries than the beam (uniform)
      n_sel = k + 1;
      if ((win.e % BIG_THREADS) == tid) {
        const int i = win.e / BIG_THREADS;
#pragma unroll
        for (int j = 0; j < BIG_EPT; ++j)
          if (j == i) { kq[j] = 0x7fffffff; ks[j] = NEG_INF; kr[j] = k; }
      }
    }
    // survivors: evaluate in full, write beam member `rank` (the previous frame's node-pool
    // stores are complete first: prefix-identity note at the top)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < BIG_EPT; ++i) {
      if (kr[i] < 0) continue;
      const int rank = kr[i];
      const BigEntry E = big_eval<CTX>(H, nb, beam, a.blank, tk, lq, tid + i * BIG_THREADS, a.cg, n_parent,
                                          n_token, a.weak_hash);
      BigHyp& Hn = hyp[cur ^ 1];
      const int slot = 1 + t * beam + rank;
      int node = E.key;
      if (E.key < 0) {
        node = slot;
        n_parent[slot] = E.par;
        n_token[slot] = E.token;
      }
      int tns = 0, tnsp = 0;
      if (E.tns_op == 1) {
        t_prev[slot] = E.tns_src; t_val[slot] = t; tns = slot; tnsp = E.tns_src;
      } else if (E.tns_op == 2) {
        t_prev[slot] = E.tnsp; t_val[slot] = t; tns = slot; tnsp = E.tnsp;
      }
      Hn.node[rank] = node; Hn.par[rank] = E.par; Hn.last[rank] = E.token;
      Hn.depth[rank] = E.depth;
      Hn.hash[rank] = E.hash; Hn.par_hash[rank] = E.parh;
      Hn.s[rank] = E.s; Hn.ns[rank] = E.ns; Hn.vs[rank] = E.vs; Hn.vns[rank] = E.vns;
      Hn.ts[rank] = E.ts; Hn.tns[rank] = tns; Hn.tnsp[rank] = tnsp;
      Hn.score[rank] = log_add2_fast(E.s, E.ns);
      Hn.vit[rank] = E.vs > E.vns ? E.vs : E.vns;
      Hn.tim[rank] = E.vs > E.vns ? E.ts : tns;
      Hn.cscore[rank] = E.cx; Hn.cstate[rank] = E.cs;
    }
    nb = n_sel;
    cur ^= 1;
    __syncthreads();  // new beam complete; tk / lq free for the next frame
  }
  // ---- emit the n-best list (as prefix_beam_kernel) --------------------------
  if (tid == 0) a.n_hyps[b] = nb;
  if (tid < beam) {
    const int64_t o = (int64_t)b * beam + tid;
    if (tid < nb) {
      const BigHyp& h = hyp[cur];
      const int L = h.depth[tid];
      a.hyp_lens[o] = L;
      a.hyp_scores[o] = CTX ? h.score[tid] + (-a.cg.node_score[h.cstate[tid]])
                            : h.score[tid];
      int* tkn = a.hyp_tokens + o * a.max_len;
      int* tm = a.hyp_times + o * a.max_len;
      int node = h.node[tid];
      for (int i = L - 1; i >= 0; --i) { tkn[i] = n_token[node]; node = n_parent[node]; }
      int n_t = 0;
      for (int y = h.tim[tid]; y != 0; y = t_prev[y]) ++n_t;
      int y = h.tim[tid];
      for (int i = n_t - 1; i >= 0; --i) { tm[i] = t_val[y]; y = t_prev[y]; }
      a.hyp_tlens[o] = n_t;
    } else {
      a.hyp_lens[o] = 0;
      a.hyp_tlens[o] = 0;
      a.hyp_scores[o] = NEG_INF;
    }
  }
}

}  // namespace


int ctc_logsoftmax_topk(const CtcRowArgs& a, hipStream_t s) {
  WN_CHECK(a.M > 0 && a.V > 0, "ctc: empty");
  WN_CHECK(a.k >= 1 && a.k <= a.V, "ctc: top-k must be in [1, vocab]");
  if (a.logp == nullptr && tune().ctc_wave != 0 && a.V <= 96 * 64 && a.k <= 16) {
    dim3 g(cdiv(a.M, 4)), t(256);
    if (a.V <= 8 * 64) hipLaunchKernelGGL(ctc_row_wave2_kernel<8>, g, t, 0, s, a);
    else if (a.V <= 72 * 64) hipLaunchKernelGGL(ctc_row_wave2_kernel<72>, g, t, 0, s, a);
    else hipLaunchKernelGGL(ctc_row_wave2_kernel<96>, g, t, 0, s, a);
    WN_HIP(hipGetLastError());
    return 0;
  }
  const size_t lds = (size_t)a.V * sizeof(float);
  WN_CHECK(lds <= 120 * 1024, "ctc: vocabulary too large for the LDS row buffer");
  WN_MAX_DYN_LDS(ctc_row_kernel, 120 * 1024);
  hipLaunchKernelGGL(ctc_row_kernel, dim3(a.M), dim3(256), lds, s, a);
  WN_HIP(hipGetLastError());
  return 0;
}

// ---- filter_blank_embedding (asr_model.py:153-180) ------------------------------------------
// One block per utterance: the frames whose CTC arg-max is not token 0, in order, as source
// row indices (map[off[b] + i], i < n_keep[b]); n_keep[b] = their number.
__global__ __launch_bounds__(256) void nonblank_map_kernel(const int* __restrict__ top1,
                                                           int stride, const int* __restrict__ off,
                                                           const int* __restrict__ len,
                                                           int* __restrict__ map,
                                                           int* __restrict__ n_keep) {
  __shared__ int wsum[4];
  __shared__ int base;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int o = off[b], n = len[b];
  if (tid == 0) base = 0;
  __syncthreads();
  for (int t0 = 0; t0 < n; t0 += 256) {
    const int t = t0 + tid;
    const bool keep = t < n && top1[(int64_t)(o + t) * stride] != 0;
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int before = base;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (keep) map[o + before + __popcll(m & ((1ull << lane) - 1))] = o + t;
    __syncthreads();
    if (tid == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (tid == 0) n_keep[b] = base;
}

// dst row (new layout: utterance b at noff[b], nlen[b] rows) <- src row map[off[b] + i] for
// i < n_keep[b], zeros behind (the reference pads the selected rows with zeros and the decoder
// attends to that padding: search.py:396 slices with the UNFILTERED lengths)
__global__ __launch_bounds__(256) void nonblank_gather_kernel(
    const float* __restrict__ src, const int* __restrict__ map, const int* __restrict__ off,
    const int* __restrict__ n_keep, const int* __restrict__ noff, const int* __restrict__ nlen,
    const int* __restrict__ row_utt_new, float* __restrict__ dst, int D4, int rows_new) {
  const int r = blockIdx.x;
  if (r >= rows_new) return;
  const int b = row_utt_new[r];
  const int i = r - noff[b];
  const f32x4* s4 = i < n_keep[b]
                        ? reinterpret_cast<const f32x4*>(src) + (int64_t)map[off[b] + i] * D4
                        : nullptr;
  f32x4* d4 = reinterpret_cast<f32x4*>(dst) + (int64_t)r * D4;
  for (int c = threadIdx.x; c < D4; c += 256) d4[c] = s4 ? s4[c] : f32x4{0.f, 0.f, 0.f, 0.f};
}

int ctc_greedy_collapse(const int* top1, int top1_stride, const int* off,
                        const int* len, int B, int blank, int* out_tokens,
                        int out_stride, int* out_lens, hipStream_t s) {
  WN_CHECK(B > 0, "greedy: empty batch");
  hipLaunchKernelGGL(greedy_kernel, dim3(B), dim3(64), 0, s, top1, top1_stride,
                     off, len, blank, out_tokens, out_stride, out_lens);
  WN_HIP(hipGetLastError());
  return 0;
}

int log_add_pairs(const double* a, const double* b, double* out, int n,
                  hipStream_t s) {
  WN_CHECK(n > 0 && a && b && out, "log_add: bad argument");
  hipLaunchKernelGGL(log_add_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, a, b, out, n);
  WN_HIP(hipGetLastError());
  return 0;
}

int64_t prefix_beam_pool_ints(int max_len, int beam) {
  return 4 * ((int64_t)max_len * beam + 1);
}


template <int NCH, bool CTX, bool LPOOL>
static int launch_pb(const PrefixBeamArgs& a, size_t pool_bytes, hipStream_t s) {
  auto kern = prefix_beam_kernel<NCH, CTX, LPOOL>;
  if (LPOOL) {
    WN_MAX_DYN_LDS(kern, PB_LPOOL_BYTES);
  }
  hipLaunchKernelGGL(kern, dim3(a.B), dim3(PB_THREADS), pool_bytes, s, a);
  return 0;
}

int ctc_prefix_beam(const PrefixBeamArgs& a_in, hipStream_t s) {
  PrefixBeamArgs a = a_in;
  a.weak_hash = tune().beam_weak_hash;
  WN_CHECK(a.B > 0, "prefix beam: empty batch");
  WN_CHECK(a.beam >= 1 && a.beam <= BIGB,
           "prefix beam: beam_size must be in [1, 64]");
  WN_CHECK(a.k == a.beam, "prefix beam: top-k width must equal the beam");
  if (a.beam > MAXB) {  // large beams: the general kernel
    if (a.cg.keys == nullptr)
      hipLaunchKernelGGL((prefix_beam_big_kernel<false>), dim3(a.B), dim3(BIG_THREADS), 0, s, a);
    else
      hipLaunchKernelGGL((prefix_beam_big_kernel<true>), dim3(a.B), dim3(BIG_THREADS), 0, s, a);
    WN_HIP(hipGetLastError());
    return 0;
  }
  static_assert(64 + MAXB * MAXB <= PB_THREADS, "one thread per entry");
  static_assert(PB_CHUNK * MAXB <= PB_THREADS, "one thread per staged top-k pair");
  const bool small = MAXB + a.beam * a.beam <= 128;
  const size_t pool_bytes = (size_t)prefix_beam_pool_ints(a.max_len, a.beam) * sizeof(int);
  const bool lpool = pool_bytes <= PB_LPOOL_BYTES && tune().beam_lds_pool != 0;
  int rc;
  if (a.cg.keys == nullptr) {
    if (small) rc = lpool ? launch_pb<2, false, true>(a, pool_bytes, s) : launch_pb<2, false, false>(a, 0, s);
    else rc = lpool ? launch_pb<PB_CHUNKS, false, true>(a, pool_bytes, s) : launch_pb<PB_CHUNKS, false, false>(a, 0, s);
  } else {
    if (small) rc = lpool ? launch_pb<2, true, true>(a, pool_bytes, s) : launch_pb<2, true, false>(a, 0, s);
    else rc = lpool ? launch_pb<PB_CHUNKS, true, true>(a, pool_bytes, s) : launch_pb<PB_CHUNKS, true, false>(a, 0, s);
  }
  if (rc != 0) return rc;
  WN_HIP(hipGetLastError());
  return 0;
}

int nonblank_map(const int* top1, int stride, const int* off, const int* len, int B, int* map,
                 int* n_keep, hipStream_t s) {
  hipLaunchKernelGGL(nonblank_map_kernel, dim3(B), dim3(256), 0, s, top1, stride, off, len, map,
                     n_keep);
  WN_HIP(hipGetLastError());
  return 0;
}

int nonblank_gather(const float* src, const int* map, const int* off, const int* n_keep,
                    const int* noff, const int* nlen, const int* row_utt_new, float* dst, int D,
                    int rows_new, hipStream_t s) {
  if (rows_new <= 0) return 0;
  hipLaunchKernelGGL(nonblank_gather_kernel, dim3(rows_new), dim3(256), 0, s, src, map, off,
                     n_keep, noff, nlen, row_utt_new, dst, D / 4, rows_new);
  WN_HIP(hipGetLastError());
  return 0;
}

}  // namespace wn
