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
  __shared__ VI chosen;
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
  for (int i = tid; i < a.V; i += 256) sm += expf(srow[i] - mx);
  sm = wave_sum(sm);
  if (lane == 0) red[4 + wave] = sm;
  __syncthreads();
  const float lsum = logf(red[4] + red[5] + red[6] + red[7]);
  if (a.logp) {
    float* o = a.logp + (int64_t)row * a.ld_out;
    for (int i = tid; i < a.V; i += 256) o[i] = (srow[i] - mx) - lsum;
  }
  VI prev;
  prev.v = INFINITY;
  prev.i = -1;
  for (int r = 0; r < a.k; ++r) {
    VI best;
    best.v = -INFINITY;
    best.i = 0x7fffffff;
    for (int i = tid; i < a.V; i += 256) {
      const float v = srow[i];
      if (v < prev.v || (v == prev.v && i > prev.i)) {
        VI c;
        c.v = v;
        c.i = i;
        best = vi_better(best, c);
      }
    }
    best = vi_wave(best);
    if (lane == 0) redvi[wave] = best;
    __syncthreads();
    if (tid == 0) {
      VI b = vi_better(vi_better(redvi[0], redvi[1]),
                       vi_better(redvi[2], redvi[3]));
      chosen = b;
      a.topk_val[(int64_t)row * a.k + r] = (b.v - mx) - lsum;
      a.topk_idx[(int64_t)row * a.k + r] = b.i;
    }
    __syncthreads();
    prev = chosen;
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

__device__ __forceinline__ double log_add2(double a, double b) {
  // wenet/utils/common.py:302-310 for two arguments
  if (a == NEG_INF && b == NEG_INF) return NEG_INF;
  const double m = a > b ? a : b;
  return m + log(exp(a - m) + exp(b - m));
}

// Prefix identity is the token sequence (the reference keys a dict by the
// tuple).  A physical node id is NOT an identity: a prefix that left the beam
// can be re-created later as a new node while its old children are still
// alive.  Every prefix therefore carries a 64-bit hash of its token sequence
// (chained splitmix64), and all "is this the same prefix / is this its parent"
// tests compare hashes.
typedef unsigned long long u64;
__device__ __forceinline__ u64 prefix_hash(u64 h, int tok) {
  u64 z = h ^ ((u64)(tok + 1) * 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
constexpr u64 ROOT_HASH = 0x243F6A8885A308D3ull;

struct Hyp {  // one beam member
  u64 hash, par_hash;
  int node, last, par;
  int ts, tns;  // heads of times_s / times_ns lists (0 = empty list)
  double s, ns, vs, vns;
  double score, vit;  // score() and viterbi_score()
  int tim;            // times() head
};

struct Entry {
  u64 hash, par_hash;
  double s, ns, vs, vns, score;
  int seq;
  int key_node;          // >=0: existing prefix node; -1: new child
  int par_node, token;   // for a new child
  int ts;                // times_s head
  int tns_src, tns_op;   // 0 empty, 1 append t, 2 replace last with t, 3 share
  int valid;
};

__global__ __launch_bounds__(256) void prefix_beam_kernel(PrefixBeamArgs a) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int T = a.len[b], off = a.off[b];
  const int beam = a.beam;
  __shared__ Hyp hyp[2][MAXB];
  __shared__ Entry ent[MAXE];
  __shared__ int tok[MAXB];
  __shared__ double lp[MAXB];
  __shared__ int s_nb, s_nvalid;

  const int cap = a.max_len * beam + 1;
  int* pool = a.pool + (int64_t)b * a.pool_stride;
  int* n_parent = pool;            // prefix nodes
  int* n_token = pool + cap;
  int* n_depth = pool + 2 * cap;
  int* t_prev = pool + 3 * cap;    // time nodes
  int* t_val = pool + 4 * cap;

  if (tid == 0) {
    n_parent[0] = -1; n_token[0] = -1; n_depth[0] = 0;
    t_prev[0] = 0; t_val[0] = -1;
    Hyp h;
    h.hash = ROOT_HASH; h.par_hash = 0;
    h.node = 0; h.last = -1; h.par = -1; h.ts = 0; h.tns = 0;
    h.s = 0.0; h.ns = NEG_INF; h.vs = 0.0; h.vns = 0.0;  // search.py:144-147
    h.score = 0.0; h.vit = 0.0; h.tim = 0;
    hyp[0][0] = h;
    s_nb = 1;
  }
  __syncthreads();

  int cur = 0;
  for (int t = 0; t < T; ++t) {
    const int nb = s_nb;
    const Hyp* H = hyp[cur];
    if (tid < beam) {
      tok[tid] = a.topk_idx[(int64_t)(off + t) * a.k + tid];
      lp[tid] = (double)a.topk_val[(int64_t)(off + t) * a.k + tid];
    }
    __syncthreads();
    const int n_ent = nb + nb * beam;
    for (int e = tid; e < n_ent; e += 256) {
      Entry E;
      E.valid = 0;
      E.s = NEG_INF; E.ns = NEG_INF; E.vs = NEG_INF; E.vns = NEG_INF;
      E.ts = 0; E.tns_src = 0; E.tns_op = 0;
      E.key_node = -1; E.par_node = -1; E.token = -1; E.seq = 0x7fffffff;
      E.hash = 0; E.par_hash = 0;
      if (e < nb) {
        // ---- unchanged prefix K = H[r] -----------------------------------
        const int r = e;
        const Hyp K = H[r];
        int qb = -1, ql = -1;
        for (int q = 0; q < beam; ++q) {
          if (tok[q] == a.blank) qb = q;
          if (K.last >= 0 && tok[q] == K.last) ql = q;
        }
        if (qb >= 0 || ql >= 0) {
          E.valid = 1;
          E.key_node = K.node; E.par_node = K.par; E.token = K.last;
          E.hash = K.hash; E.par_hash = K.par_hash;
          int seq = 0x7fffffff;
          if (qb >= 0) {
            const double p = lp[qb];
            E.s = K.score + p;       // log_add(-inf, x) == x
            E.vs = K.vit + p;
            E.ts = K.tim;
            seq = min(seq, (qb * nb + r) * 2);
          }
          if (ql >= 0) {
            const double p = lp[ql];
            const int u = K.last;
            int rp = -1;
            for (int j = 0; j < nb; ++j)
              if (H[j].hash == K.par_hash) rp = j;
            const double xa = K.ns + p, va = K.vns + p;
            seq = min(seq, (ql * nb + r) * 2);
            double v = NEG_INF, ctp = NEG_INF;
            int tsrc = 0, top = 0;
            if (rp < 0) {
              E.ns = xa;
              if (v < va) { v = va; tsrc = K.tns; top = 2; }
            } else {
              const Hyp P = H[rp];
              double xb, vb; int tb, sub;
              if (P.last == u) { xb = P.s + p; vb = P.vs + p; tb = P.ts; sub = 1; }
              else { xb = P.score + p; vb = P.vit + p; tb = P.tim; sub = 0; }
              seq = min(seq, (ql * nb + rp) * 2 + sub);
              if (r < rp) {          // hyp K is visited before its parent
                E.ns = log_add2(xa, xb);
                if (v < va) { v = va; ctp = p; tsrc = K.tns; top = 2; }
                if (v < vb) { v = vb; ctp = p; tsrc = tb; top = 1; }
              } else {               // parent first
                E.ns = log_add2(xb, xa);
                if (v < vb) { v = vb; ctp = p; tsrc = tb; top = 1; }
                if (v < va) {
                  v = va;
                  if (ctp < p) { ctp = p; tsrc = K.tns; top = 2; }
                }
              }
            }
            E.vns = v; E.tns_src = tsrc; E.tns_op = top;
          }
          E.seq = seq;
        }
      } else {
        // ---- extension P + u ---------------------------------------------
        const int r = (e - nb) / beam, q = (e - nb) % beam;
        const Hyp P = H[r];
        const int u = tok[q];
        const u64 ch = prefix_hash(P.hash, u);
        bool merged = false;
        for (int j = 0; j < nb; ++j)
          if (H[j].hash == ch) merged = true;
        if (u != a.blank && !merged) {
          const double p = lp[q];
          double x, v; int tb, sub;
          if (u == P.last) { x = P.s + p; v = P.vs + p; tb = P.ts; sub = 1; }
          else { x = P.score + p; v = P.vit + p; tb = P.tim; sub = 0; }
          E.valid = 1;
          E.ns = x;
          if (v > NEG_INF) { E.vns = v; E.tns_src = tb; E.tns_op = 1; }
          E.key_node = -1; E.par_node = P.node; E.token = u;
          E.hash = ch; E.par_hash = P.hash;
          E.seq = (q * nb + r) * 2 + sub;
        }
      }
      E.score = log_add2(E.s, E.ns);
      ent[e] = E;
    }
    __syncthreads();
    // ---- second beam prune: stable sort by score desc, keep `beam` -----------
    if (tid == 0) s_nvalid = 0;
    __syncthreads();
    Hyp* Hn = hyp[cur ^ 1];
    for (int e = tid; e < n_ent; e += 256) {
      const Entry E = ent[e];
      if (!E.valid) continue;
      atomicAdd(&s_nvalid, 1);
      int rank = 0;
      for (int j = 0; j < n_ent; ++j) {
        if (!ent[j].valid) continue;
        const double sj = ent[j].score;
        if (sj > E.score || (sj == E.score && ent[j].seq < E.seq)) ++rank;
      }
      if (rank < beam) {
        Hyp h;
        const int slot = 1 + t * beam + rank;
        if (E.key_node >= 0) {
          h.node = E.key_node;
        } else {
          h.node = slot;
          n_parent[slot] = E.par_node;
          n_token[slot] = E.token;
          n_depth[slot] = n_depth[E.par_node] + 1;
        }
        h.par = E.par_node; h.last = E.token;
        h.hash = E.hash; h.par_hash = E.par_hash;
        h.s = E.s; h.ns = E.ns; h.vs = E.vs; h.vns = E.vns;
        h.ts = E.ts;
        if (E.tns_op == 1) {
          t_prev[slot] = E.tns_src; t_val[slot] = t; h.tns = slot;
        } else if (E.tns_op == 2) {
          t_prev[slot] = t_prev[E.tns_src]; t_val[slot] = t; h.tns = slot;
        } else {
          h.tns = 0;
        }
        h.score = E.score;
        h.vit = h.vs > h.vns ? h.vs : h.vns;   // search.py:87-88
        h.tim = h.vs > h.vns ? h.ts : h.tns;   // search.py:90-91
        Hn[rank] = h;
      }
    }
    __threadfence_block();
    __syncthreads();
    if (tid == 0) s_nb = min(beam, s_nvalid);
    cur ^= 1;
    __syncthreads();
  }

  // ---- emit the n-best list -------------------------------------------------
  const int nb = s_nb;
  if (tid == 0) a.n_hyps[b] = nb;
  if (tid < beam) {
    const int64_t o = (int64_t)b * beam + tid;
    if (tid < nb) {
      const Hyp h = hyp[cur][tid];
      const int L = n_depth[h.node];
      a.hyp_lens[o] = L;
      a.hyp_scores[o] = h.score;
      int* tk = a.hyp_tokens + o * a.max_len;
      int node = h.node;
      for (int i = L - 1; i >= 0; --i) { tk[i] = n_token[node]; node = n_parent[node]; }
      int n_t = 0;
      for (int x = h.tim; x != 0; x = t_prev[x]) ++n_t;
      int* tm = a.hyp_times + o * a.max_len;
      int x = h.tim;
      for (int i = n_t - 1; i >= 0; --i) { tm[i] = t_val[x]; x = t_prev[x]; }
      a.hyp_tlens[o] = n_t;  // == L, or 0 for a never-set list
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
  const size_t lds = (size_t)a.V * sizeof(float);
  WN_CHECK(lds <= 120 * 1024, "ctc: vocabulary too large for the LDS row buffer");
  static size_t attr = 0;
  if (lds > attr) {
    WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ctc_row_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
    attr = lds;
  }
  hipLaunchKernelGGL(ctc_row_kernel, dim3(a.M), dim3(256), lds, s, a);
  WN_HIP(hipGetLastError());
  return 0;
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

int64_t prefix_beam_pool_ints(int max_len, int beam) {
  return 5 * ((int64_t)max_len * beam + 1);
}

int ctc_prefix_beam(const PrefixBeamArgs& a, hipStream_t s) {
  WN_CHECK(a.B > 0, "prefix beam: empty batch");
  WN_CHECK(a.beam >= 1 && a.beam <= MAXB,
           "prefix beam: beam_size must be in [1, 16]");
  WN_CHECK(a.k == a.beam, "prefix beam: top-k width must equal the beam");
  hipLaunchKernelGGL(prefix_beam_kernel, dim3(a.B), dim3(256), 0, s, a);
  WN_HIP(hipGetLastError());
  return 0;
}

}  // namespace wn
