// Device side of the `attention` decode mode -- attention_beam_search,
// wenet/models/transformer/search.py:252-371 with TransformerDecoder.forward_one_step,
// decoder.py:226-281 -- as this path runs it:
//
//  * every running hypothesis owns ONE row per step: the decoder is evaluated on the
//    newest token only; the self-attention keys / values of earlier positions come
//    from a per-layer cache [step][slot][K | V].  A hypothesis does not keep its own
//    copy of the cache: `path[slot][j]` names the slot whose row holds position j of
//    its prefix (the ancestor that was alive at step j), so a beam update copies B*N
//    short index rows instead of re-gathering B*N x L x 2d floats per layer;
//  * self_attn_step_kernel: one wave per (hypothesis, head), scores of up to 64 cached
//    keys at a time (lane = key), online softmax across blocks, then P.V with
//    lane = dimension (coalesced cache rows);
//  * beam_update_kernel: the N x N candidates of an utterance (mask_finished_scores /
//    mask_finished_preds of utils/mask.py:258-304 applied), ranked (value descending,
//    flat index ascending) by counting, the best N written in that order -- what
//    `scores.view(B, N*N).topk(N)` returns -- together with the children's token rows,
//    path rows and end flags;
//  * beam_finish_kernel: length penalty, arg-max, the winner without <sos> / <eos>.
// All scores are fp32 like the reference's score tensors.
#include <algorithm>

#include "kernels.h"

namespace wn {

namespace {

// q: [n][ldq] (this step's Q at column h*64); cache: [steps][n][2d] (K at h*64, V at
// d + h*64); path: [n][max_len]; len = number of positions (the newest included)
__global__ __launch_bounds__(64) void self_attn_step_kernel(
    const float* __restrict__ q, int ldq, const float* __restrict__ cache, int n, int d,
    const int* __restrict__ path, int max_len, int len, float scale, float* __restrict__ out,
    int ldo) {
  const int r = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
  const float* qp = q + (int64_t)r * ldq + h * 64;
  f32x4 qv[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) qv[i] = *reinterpret_cast<const f32x4*>(qp + 4 * i);
  float m_run = -INFINITY, l_run = 0.f, acc = 0.f;   // acc: output dim `lane`
  for (int j0 = 0; j0 < len; j0 += 64) {
    const int j = j0 + lane;
    float sc = -INFINITY;
    int row = 0;
    if (j < len) {
      row = j * n + path[(int64_t)r * max_len + j];
      const float* kp = cache + (int64_t)row * 2 * d + h * 64;
      float dot = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const f32x4 kv = *reinterpret_cast<const f32x4*>(kp + 4 * i);
        dot += qv[i][0] * kv[0] + qv[i][1] * kv[1] + qv[i][2] * kv[2] + qv[i][3] * kv[3];
      }
      sc = dot * scale;
    }
    const float m_new = fmaxf(m_run, wave_max(sc));
    const float p = j < len ? __expf(sc - m_new) : 0.f;
    const float corr = __expf(m_run - m_new);
    l_run = l_run * corr + wave_sum(p);
    acc *= corr;
    const int nb = min(64, len - j0);
    for (int t = 0; t < nb; ++t) {
      const float pt = __shfl(p, t, 64);
      const int rt = __shfl(row, t, 64);
      acc += pt * cache[(int64_t)rt * 2 * d + d + h * 64 + lane];
    }
    m_run = m_new;
  }
  out[(int64_t)r * ldo + h * 64 + lane] = acc / l_run;
}

// K | V of this step's rows -> cache[step]
__global__ void cache_store_kernel(const float* __restrict__ qkv, int d, int n,
                                   float* __restrict__ cache_step) {
  const int r = blockIdx.x;
  const f32x4* s = reinterpret_cast<const f32x4*>(qkv + (int64_t)r * 3 * d + d);
  f32x4* o = reinterpret_cast<f32x4*>(cache_step + (int64_t)r * 2 * d);
  for (int i = threadIdx.x; i < 2 * d / 4; i += blockDim.x) o[i] = s[i];
}

// One block per utterance, the N*N candidates strided over its threads (N <= 64).  `step` =
// length of the parents' token rows (the new token lands at index `step`).  A token index
// outside [0, V) -- the top-k of a row of NaN / -inf logits has no valid entry -- ends the
// hypothesis (eos) instead of indexing the embedding table with it.
__global__ void beam_update_kernel(int N, int step, int max_len, int eos, int V,
                                   const float* __restrict__ topv, const int* __restrict__ topi,
                                   const float* __restrict__ score_in, const int* __restrict__ end_in,
                                   const int* __restrict__ tok_in, const int* __restrict__ path_in,
                                   float* __restrict__ score_out, int* __restrict__ end_out,
                                   int* __restrict__ tok_out, int* __restrict__ path_out,
                                   int* __restrict__ last_tok, int* __restrict__ n_running_done) {
  extern __shared__ float cand[];          // [N*N] values, then N ints (winner flat index)
  int* win = reinterpret_cast<int*>(cand + N * N);
  const int b = blockIdx.x;
  const int nn = N * N;
  for (int t = threadIdx.x; t < nn; t += blockDim.x) {
    const int n = t / N, k = t % N;
    const int hyp = b * N + n;
    float lp = topv[(int64_t)hyp * N + k];
    if (end_in[hyp]) lp = k == 0 ? 0.f : -INFINITY;     // mask_finished_scores
    cand[t] = score_in[hyp] + lp;
  }
  __syncthreads();
  // rank by (value descending, flat index ascending): what scores.view(B, N*N).topk(N) returns
  for (int t = threadIdx.x; t < nn; t += blockDim.x) {
    const float v = cand[t];
    int rank = 0;
    for (int u = 0; u < nn; ++u) {
      const float o = cand[u];
      rank += (o > v) || (o == v && u < t);
    }
    if (rank < N) win[rank] = t;
  }
  __syncthreads();
  // child c of this utterance <- candidate win[c]
  for (int c = threadIdx.x; c < N; c += blockDim.x) {
    const int f = win[c];
    const int n = f / N, k = f % N;
    const int parent = b * N + n, child = b * N + c;
    int tok = end_in[parent] ? eos : topi[(int64_t)parent * N + k];  // mask_finished_preds
    if ((unsigned)tok >= (unsigned)V) tok = eos;
    score_out[child] = cand[f];
    for (int j = 0; j < step; ++j) {
      tok_out[(int64_t)child * max_len + j] = tok_in[(int64_t)parent * max_len + j];
      if (j + 1 < step) path_out[(int64_t)child * max_len + j] = path_in[(int64_t)parent * max_len + j];
    }
    // position step-1 (the parent's newest token) was computed in the parent's row
    path_out[(int64_t)child * max_len + step - 1] = parent;
    path_out[(int64_t)child * max_len + step] = child;   // its own row at the next step
    tok_out[(int64_t)child * max_len + step] = tok;
    last_tok[child] = tok;
    const int ended = tok == eos;
    end_out[child] = ended;
    if (ended) atomicAdd(n_running_done, 1);
  }
}

__global__ void beam_finish_kernel(int N, int len, int max_len, int eos, float length_penalty,
                                   const float* __restrict__ score, const int* __restrict__ tok,
                                   int* __restrict__ out_tok, int* __restrict__ out_len) {
  const int b = blockIdx.x;
  if (threadIdx.x != 0) return;
  int best = 0;
  float best_s = -INFINITY;
  for (int n = 0; n < N; ++n) {
    const int* row = tok + (int64_t)(b * N + n) * max_len;
    int cnt = 0;
    for (int j = 0; j < len; ++j) cnt += row[j] != eos;      // hyps.ne(eos).sum(1)
    const float s = score[b * N + n] / powf((float)cnt, length_penalty);
    if (s > best_s) { best_s = s; best = n; }                // first maximum, like max()
  }
  const int* row = tok + (int64_t)(b * N + best) * max_len;
  int o = 0;
  for (int j = 1; j < len; ++j)
    if (row[j] != eos) out_tok[(int64_t)b * max_len + o++] = row[j];
  out_len[b] = o;
}

// x[r] = embed[last_tok[r]] * sqrt(d) + pe[pos]  (embedding.py:58-76 for the newest token)
__global__ void step_embed_kernel(const int* __restrict__ last_tok, int pos,
                                  const float* __restrict__ emb, const float* __restrict__ pe,
                                  float scale, int D4, float* __restrict__ x) {
  const int r = blockIdx.x;
  const f32x4* e = reinterpret_cast<const f32x4*>(emb + (int64_t)last_tok[r] * D4 * 4);
  const f32x4* p = reinterpret_cast<const f32x4*>(pe + (int64_t)pos * D4 * 4);
  f32x4* o = reinterpret_cast<f32x4*>(x + (int64_t)r * D4 * 4);
  for (int i = threadIdx.x; i < D4; i += blockDim.x) o[i] = e[i] * scale + p[i];
}

__global__ void beam_init_kernel(int BN, int N, int max_len, int sos, float* score, int* end,
                                 int* tok, int* path, int* last_tok) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= BN) return;
  score[r] = (r % N) == 0 ? 0.f : -INFINITY;
  end[r] = 0;
  tok[(int64_t)r * max_len] = sos;
  path[(int64_t)r * max_len] = r;                 // global row of position 0
  last_tok[r] = sos;
}

}  // namespace

int attn_self_step(const float* qkv, int d, int heads, int n, float* cache, int step,
                   const int* path, int max_len, float* out, hipStream_t s) {
  hipLaunchKernelGGL(cache_store_kernel, dim3(n), dim3(128), 0, s, qkv, d, n,
                     cache + (int64_t)step * n * 2 * d);
  hipLaunchKernelGGL(self_attn_step_kernel, dim3(n, heads), dim3(64), 0, s, qkv, 3 * d, cache, n,
                     d, path, max_len, step + 1, 0.125f, out, d);
  WN_HIP(hipGetLastError());
  return 0;
}

int attn_step_embed(const int* last_tok, int pos, const float* emb, const float* pe,
                    float scale, int d, int n, float* x, hipStream_t s) {
  hipLaunchKernelGGL(step_embed_kernel, dim3(n), dim3(64), 0, s, last_tok, pos, emb, pe, scale,
                     d / 4, x);
  WN_HIP(hipGetLastError());
  return 0;
}

int attn_beam_init(int BN, int N, int max_len, int sos, float* score, int* end, int* tok,
                   int* path, int* last_tok, hipStream_t s) {
  hipLaunchKernelGGL(beam_init_kernel, dim3(cdiv(BN, 256)), dim3(256), 0, s, BN, N, max_len, sos,
                     score, end, tok, path, last_tok);
  WN_HIP(hipGetLastError());
  return 0;
}

int attn_beam_update(int B, int N, int step, int max_len, int eos, int V, const float* topv,
                     const int* topi, const float* score_in, const int* end_in,
                     const int* tok_in, const int* path_in, float* score_out, int* end_out,
                     int* tok_out, int* path_out, int* last_tok, int* n_done, hipStream_t s) {
  WN_CHECK(N >= 1 && N <= 64, "attention beam search: beam_size must be in [1, 64]");
  const int thr = std::min(1024, (N * N + 63) / 64 * 64);
  hipLaunchKernelGGL(beam_update_kernel, dim3(B), dim3(thr), (N * N + N) * sizeof(float), s, N,
                     step, max_len, eos, V, topv, topi, score_in, end_in, tok_in, path_in,
                     score_out, end_out, tok_out, path_out, last_tok, n_done);
  WN_HIP(hipGetLastError());
  return 0;
}

int attn_beam_finish(int B, int N, int len, int max_len, int eos, float length_penalty,
                     const float* score, const int* tok, int* out_tok, int* out_len,
                     hipStream_t s) {
  hipLaunchKernelGGL(beam_finish_kernel, dim3(B), dim3(64), 0, s, N, len, max_len, eos,
                     length_penalty, score, tok, out_tok, out_len);
  WN_HIP(hipGetLastError());
  return 0;
}

}  // namespace wn
