// Internal state of libwenet_amd shared by the engine (model.hip: launch sequences of the
// encoder / decoders, weight images) and the C ABI (cabi.hip: include/wenet_amd.h entry points):
// device buffers, the re-laid-out weight views, the per-handle workspace `wn_model`, the
// per-call guards (one host thread per handle, operand precision of the calling thread).
#pragma once
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/wenet_amd.h"
#include "kernels.h"

namespace wn {

const char* last_error_cstr();


// ---------------------------------------------------------------------------
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;             // owns its allocation
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) WN_HIP(hipFree(p));
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    WN_HIP(hipMalloc(&p, want));
    cap = want;
    return 0;
  }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Pinned host staging for the small per-call descriptor uploads.  The event
// makes the next call wait only for the previous call's H2D copies.
struct Stager {
  char* host = nullptr;
  size_t cap = 0, used = 0;
  hipEvent_t ev = nullptr;
  bool pending = false;
  Stager() = default;
  Stager(const Stager&) = delete;
  Stager& operator=(const Stager&) = delete;
  ~Stager() {
    if (host) (void)hipHostFree(host);
    if (ev) (void)hipEventDestroy(ev);
  }
  int begin(size_t need) {
    if (!ev) WN_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    if (pending) { WN_HIP(hipEventSynchronize(ev)); pending = false; }
    if (need > cap) {
      if (host) WN_HIP(hipHostFree(host));
      host = nullptr;
      cap = need + need / 4 + 4096;
      WN_HIP(hipHostMalloc((void**)&host, cap, hipHostMallocDefault));
    }
    used = 0;
    return 0;
  }
  int put(DevBuf& buf, const void* data, size_t bytes, hipStream_t s) {
    WN_TRY(buf.ensure(std::max<size_t>(bytes, 16)));
    if (bytes == 0) return 0;
    const size_t o = (used + 63) / 64 * 64;
    WN_CHECK(o + bytes <= cap, "descriptor staging overflow");
    memcpy(host + o, data, bytes);
    used = o + bytes;
    WN_HIP(hipMemcpyAsync(buf.p, host + o, bytes, hipMemcpyHostToDevice, s));
    return 0;
  }
  // the same staged copy into a window of a buffer the caller sized
  int put_at(void* dst, const void* data, size_t bytes, hipStream_t s) {
    if (bytes == 0) return 0;
    const size_t o = (used + 63) / 64 * 64;
    WN_CHECK(o + bytes <= cap, "descriptor staging overflow");
    memcpy(host + o, data, bytes);
    used = o + bytes;
    WN_HIP(hipMemcpyAsync(dst, host + o, bytes, hipMemcpyHostToDevice, s));
    return 0;
  }
  int end(hipStream_t s) {
    WN_HIP(hipEventRecord(ev, s));
    pending = true;
    return 0;
  }
};

// Pinned host memory that only grows: the landing area of result copies (ONE device -> host
// copy per search instead of one staged copy per pageable result array).
struct PinnedBuf {
  char* p = nullptr;
  size_t cap = 0;
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf&) = delete;
  PinnedBuf& operator=(const PinnedBuf&) = delete;
  ~PinnedBuf() { if (p) (void)hipHostFree(p); }
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) WN_HIP(hipHostFree(p));
    p = nullptr;
    cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    WN_HIP(hipHostMalloc((void**)&p, want, hipHostMallocDefault));
    cap = want;
    return 0;
  }
};

// A second stream of the handle + the two events that order it with the caller's stream: work
// that does not depend on the search result runs there while the (latency-bound, few-CU) prefix
// beam search runs on the caller's stream -- wn_rescore_prefetch in cabi.hip.
struct SideStream {
  hipStream_t st = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  SideStream() = default;
  SideStream(const SideStream&) = delete;
  SideStream& operator=(const SideStream&) = delete;
  int ensure() {
    if (st) return 0;
    WN_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    WN_HIP(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
    WN_HIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    return 0;
  }
  ~SideStream() {
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (st) (void)hipStreamDestroy(st);
  }
};

struct Linear { const float* w = nullptr; const float* b = nullptr; int out = 0, in = 0; };
struct Norm { const float* w = nullptr; const float* b = nullptr; };

struct EncLayer {
  Norm norm_ff_mac, norm_mha, norm_conv, norm_ff, norm_final, conv_norm;
  Linear ffm1, ffm2, ff1, ff2, qkv, out, pw1, pw2;
  const float* bias_u = nullptr; const float* bias_v = nullptr;
  const float* pos_w = nullptr;   // linear_pos.weight [d][d]
  float* pos_tab = nullptr;       // [max_pos][d] = linear_pos(pe)
  const float* dw_wt = nullptr;   // [K][d]
  const float* dw_b = nullptr;
  const float* cpad = nullptr;    // [d]
};

struct TfLayer {  // TransformerEncoderLayer (encoder_layer.py:28-127)
  Norm n1, n2;
  Linear qkv, out, ff1, ff2;
};

struct DecLayer {
  Norm n1, n2, n3;
  Linear self_qkv, self_out, src_q, src_kv, src_out, ff1, ff2;
};

struct Decoder {
  const float* embed = nullptr;  // [V][d]
  const float* pe = nullptr;     // [max_pos][d]
  Norm after;
  Linear out;
  std::vector<DecLayer> layers;
};

namespace {    // small kernels: a private copy per translation unit (-fno-gpu-rdc)


// x6 conv2: base pixel (plane image row of conv1's output, even-first order inside a
// frame) of GEMM row (g, f2): frame off1[u] + 2 t2, position f2 (= f1 2 f2)
// (fstep 1: plane image with the even f1 first; 2: the plain channels-last tensor)
__global__ void build_conv2_pix_kernel(const int* row_utt2, const int* off2, const int* off1,
                                       int M, int F1, int F2, int fstep, int* a_pix) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * F2) return;
  const int g = i / F2, f2 = i % F2;
  const int u = row_utt2[g];
  a_pix[i] = (off1[u] + 2 * (g - off2[u])) * F1 + fstep * f2;
}

__global__ void build_conv2_rows_kernel(const int* row_utt2, const int* off2,
                                        const int* off1, int M, int F1, int F2,
                                        int C, int64_t* a_row_off) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * F2) return;
  const int g = i / F2, f2 = i % F2;
  const int u = row_utt2[g];
  const int t2 = g - off2[u];
  const int64_t t1 = off1[u] + 2 * t2;
  a_row_off[i] = (t1 * F1 + 2 * f2) * (int64_t)C;
}

// packed rows -> padded (B, Tp, D) with zero fill
__global__ void scatter_padded_kernel(const float* src, int lds, const int* off,
                                      const int* len, int Tp, int D4,
                                      float* dst) {
  const int b = blockIdx.y, t = blockIdx.x;
  f32x4* d = reinterpret_cast<f32x4*>(dst + ((int64_t)b * Tp + t) * D4 * 4);
  if (t < len[b]) {
    const f32x4* s =
        reinterpret_cast<const f32x4*>(src + (int64_t)(off[b] + t) * lds);
    for (int i = threadIdx.x; i < D4; i += blockDim.x) d[i] = s[i];
  } else {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < D4; i += blockDim.x) d[i] = z;
  }
}

// generic (non multiple-of-4 width) variant used for the (B,Tp,V) log-probs
__global__ void scatter_padded_any_kernel(const float* src, int lds,
                                          const int* off, const int* len,
                                          int Tp, int D, float* dst) {
  const int b = blockIdx.y, t = blockIdx.x;
  float* d = dst + ((int64_t)b * Tp + t) * D;
  if (t < len[b]) {
    const float* s = src + (int64_t)(off[b] + t) * lds;
    for (int i = threadIdx.x; i < D; i += blockDim.x) d[i] = s[i];
  } else {
    for (int i = threadIdx.x; i < D; i += blockDim.x) d[i] = 0.f;
  }
}

// Conv1dSubsampling2 front end: utterance b becomes the packed segment
// [0, x_0 .. x_{len-1}, 0, 0] (len + 3 rows of F floats) so that the k=3, pad=1
// convolution over time is a plain GEMM over three consecutive rows.
__global__ void pad_feats_kernel(const float* feats, int T, int F, const int* seg_off,
                                 const int* len, const float* mean,
                                 const float* istd, float* xpad) {
  const int b = blockIdx.y, j = blockIdx.x;
  const int L = len[b];
  if (j >= L + 3) return;
  float* dst = xpad + (int64_t)(seg_off[b] + j) * F;
  const int t = j - 1;
  if (t >= 0 && t < L) {
    const float* src = feats + ((int64_t)b * T + t) * F;
    for (int i = threadIdx.x; i < F; i += blockDim.x) {
      float v = src[i];
      if (mean) v = (v - mean[i]) * istd[i];
      dst[i] = v;
    }
  } else {
    for (int i = threadIdx.x; i < F; i += blockDim.x) dst[i] = 0.f;
  }
}

__global__ void zero_rows_kernel(float* base, int D4, const int* rows, int n) {
  const int r = blockIdx.x;
  if (r >= n) return;
  f32x4* d = reinterpret_cast<f32x4*>(base + (int64_t)rows[r] * D4 * 4);
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < D4; i += blockDim.x) d[i] = z;
}

__global__ void embed_kernel(const int* tok, const int* pos, const float* emb,
                             const float* pe, float scale, int D4, float* x) {
  const int r = blockIdx.x;
  const f32x4* e = reinterpret_cast<const f32x4*>(emb + (int64_t)tok[r] * D4 * 4);
  const f32x4* p = reinterpret_cast<const f32x4*>(pe + (int64_t)pos[r] * D4 * 4);
  f32x4* o = reinterpret_cast<f32x4*>(x + (int64_t)r * D4 * 4);
  for (int i = threadIdx.x; i < D4; i += blockDim.x) o[i] = e[i] * scale + p[i];
}

// log_softmax(row)[target] -- forward_attention_decoder's log_softmax
// (asr_model.py:541-546) fused with the gather of search.py:431-441.
__global__ __launch_bounds__(256) void row_logp_at_kernel(
    const float* logits, int ld, int V, const int* target, float* out) {
  __shared__ float red[8];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* x = logits + (int64_t)row * ld;
  float mx = -INFINITY;
  for (int i = tid; i < V; i += 256) mx = fmaxf(mx, x[i]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sm = 0.f;
  for (int i = tid; i < V; i += 256) sm += expf(x[i] - mx);
  sm = wave_sum(sm);
  if (lane == 0) red[4 + wave] = sm;
  __syncthreads();
  if (tid == 0)
    out[row] = (x[target[row]] - mx) - logf(red[4] + red[5] + red[6] + red[7]);
}

}  // namespace
}  // namespace wn

using namespace wn;


// ===========================================================================
struct wn_model {
  wn_config cfg;
  int device = 0;
  // immutable after create, shared by wn_model_clone()d handles
  std::shared_ptr<DevBuf> weights = std::make_shared<DevBuf>();  // one slab for every weight
  int64_t n_weight_elems = 0;            // floats in the slab
  std::shared_ptr<DevBuf> weights_bf16;  // bf16 image of the slab (bf16 mode, lazily)
  // MXFP8 images of the FFN weights (fp8 mode, lazily): fp32 weight pointer ->
  // (e4m3 [N][K], block scales [K/128][N] dwords); clones share it
  struct MxW { const void* q; const unsigned* scale; };
  std::shared_ptr<DevBuf> weights_mx;
  std::shared_ptr<std::map<const float*, MxW>> mx_at;
  bool fp8_ffn = false;                  // WN_PREC_FP8: prec == PREC_BF16 + MXFP8 FFN GEMMs
  // plane images of the weights the six-product fp32 GEMM runs (gemm_x6.hip): fp32 weight
  // pointer -> X3 image; built at create, shared by clones
  std::shared_ptr<DevBuf> weights_x6;
  std::shared_ptr<std::map<const float*, const void*>> x6_at;
  DevBuf nb_map, nb_keep, nb_enc, nb_off_old;   // filter_blank_embedding scratch
  std::shared_ptr<DevBuf> weights_x6p;      // k-slot-permuted FFN w_2 images (ffn_x6f.hip)
  std::shared_ptr<std::map<const float*, const void*>> x6p_at;
  // QKV weights with the rows permuted per head ([Q_h | K_h | V_h] x 64) as X3 images + the
  // biases in the same order: the QKV projection that writes the attention's key-tile images
  // itself (gemm_x6r.hip epi 4); fp32 weight pointer -> (image, bias)
  std::shared_ptr<DevBuf> weights_x6q;
  std::shared_ptr<std::map<const float*, std::pair<const void*, const float*>>> x6q_at;
  DevBuf x6_a, x6_h;                     // images of the GEMM input rows / the FFN hidden tensor
  DevBuf x6_lin;                         // image of linear()'s A operand (large fp32 GEMMs)
  // biases of the vocabulary-sized layers (CTC head, decoder output layers) padded with zeros
  // to a multiple of 4 columns: weight pointer -> padded bias
  std::shared_ptr<DevBuf> bias4_buf;
  std::shared_ptr<std::map<const float*, const float*>> bias4;
  DevBuf mx_sa, mx_sh;                   // block scales of the LN output / FFN hidden
  std::map<std::string, const float*> w; // name -> device pointer
  // re-laid-out subsampling weights
  const float* conv1_w = nullptr; const float* conv1_b = nullptr;
  Linear conv2, sub_out;
  const float* cmvn_mean = nullptr; const float* cmvn_istd = nullptr;
  const float* pe = nullptr;
  Norm after_norm;
  Linear ctc;
  std::vector<EncLayer> layers;
  std::vector<TfLayer> tf_layers;       // encoder_type 1
  Linear tconv1, tconv2;                // Conv1dSubsampling2 as gathered-row GEMMs
  bool fbank_ok = true;
  Decoder left, right;
  std::shared_ptr<DevBuf> pos_tabs = std::make_shared<DevBuf>();

  // ---- current batch ----------------------------------------------------
  int B = 0, Tp = 0, rows = 0;          // rows of the encoder-output layout
  std::vector<int> off, len;            // per utterance (rows layout)
  DevBuf d_off, d_len, d_row_utt, d_off1, d_len1, d_a_row_off;
  DevBuf c1, c2, x, t1, t2, hbuf, qkv, enc;
  DevBuf ffn_part;                      // hidden-slice partials of the fused FFN
  DevBuf attn_kbias;                    // per-key score term of the folded rel-pos attention
  DevBuf xpad, pos_rows, d_row_t, d_zero_rows;
  DevBuf ck_kv, ck_xext, ck_glu, ck_desc, ck_rowutt, ck_sess;  // forward_chunk scratch
  // Whisper log-mel: DFT / window tables (shared), mel matrix per bin count
  std::shared_ptr<DevBuf> lm_dft = std::make_shared<DevBuf>();
  // resampler taps per (orig, new) rate pair (wn_resample)
  std::shared_ptr<std::map<std::pair<int, int>, std::shared_ptr<DevBuf>>> rs_taps =
      std::make_shared<std::map<std::pair<int, int>, std::shared_ptr<DevBuf>>>();
  std::shared_ptr<std::map<int, std::shared_ptr<DevBuf>>> lm_mel =
      std::make_shared<std::map<int, std::shared_ptr<DevBuf>>>();
  DevBuf lm_off, lm_foff, lm_nfr, lm_rowutt, lm_frames, lm_spec, lm_pw, lm_melout, lm_umax;
  // ctc
  int ctc_rows = 0, ctc_k = 0;
  bool ctc_valid = false;
  bool ln0_done = false;   // t1 already holds layer 0's norm_ff_macaron(x) (sub_out_linear)
  DevBuf logits, topk_val, topk_idx;
  // searches
  DevBuf pb_dbg;
  DevBuf g_tok, g_len, pb_pool;
  DevBuf pb_out;          // the prefix beam search's results, one block: counts | lengths | scores | tokens | times
  PinnedBuf pb_host;      // ... and where they land on the host
  // the n-best of the last wn_ctc_prefix_beam_search, still in pb_out / pb_host, for wn_rescore:
  // batch size, beam, row pitch and the byte offsets of the fields inside the block
  bool pb_valid = false;
  int pb_B = 0, pb_beam = 0, pb_max_len = 0;
  size_t pb_o_sc = 0, pb_o_nh = 0, pb_o_len = 0, pb_o_tok = 0;
  // rescoring
  DevBuf r_tok, r_rtok, r_pos, r_tgt, r_rtgt, r_qoff, r_qlen, r_kvoff, r_kvlen;
  DevBuf r_x, r_t1, r_t2, r_qkv, r_h, r_mem, r_logits, r_out;
  DevBuf r_mem_all;            // per-layer cross-attention K/V of the current batch
  DevBuf r_seqsrc, r_seqfirst; // wn_rescore: sequence -> (utt, hyp) slot, first sequence per utterance
  DevBuf r_gqoff, r_gqlen, r_gkvoff, r_gkvlen;   // wn_rescore: cross-attention groups (one per utterance)
  DevBuf r_hyp;                // wn_rescore: host-given n-best (tokens | ctc scores) on the device
  DevBuf r_res;                // wn_rescore: results, one block
  PinnedBuf r_host;            // ... and where they land on the host
  DevBuf ab_cache, ab_state;   // `attention` mode: self-attention K|V cache, beam state
  bool mem_cache_valid = false;

  SideStream side;             // wn_rescore_prefetch
  // cross-attention K | V of the current batch for every decoder layer (left, then right),
  // projected ahead of the rescoring pass (wn_rescore_prefetch), and the plane image of the
  // encoder output they were projected from
  DevBuf r_kv_all, r_enc3;
  // X3 plane image of t1 = LN(x) for the fused six-product FFN (tune().ffn_ximg): valid while
  // t1_img_ok (set by the producer launch, cleared by the consumer)
  DevBuf attn_img;      // key-tile images of the six-product attention (attention_x6.hip)
  int attn_blk_off = 0, attn_n_blk = 0;   // its block list inside d_row_utt (set_layout)
  DevBuf t1_img;
  bool t1_img_ok = false;
  bool kv_ready = false;
  // wn_model_set_encode_gate: one-shot event the next wn_encode waits for.  Where: behind its
  // descriptor uploads and in front of conv1 (tune enc_gate_pos = 0, default), or behind
  // CMVN + conv1 (= 1: chained encoders of several handles overlap that HBM-bound kernel);
  // conv paths that never reach either position wait once conv2 is queued (model.hip).  The
  // gate orders work for performance only -- each handle has its own workspace, no result
  // depends on it
  hipEvent_t enc_gate = nullptr;
  int kv_rows = 0, kv_nl = 0, kv_nr = 0;
  Stager stage;
  // optional HIP-event bracket around the FFN w_1 GEMM launches (the kernel
  // the roofline is quoted on); see wn_profile_*.
  std::vector<hipEvent_t> prof_ev;
  size_t prof_used = 0;
  bool prof_on = false;
  unsigned prof_seq = 0;
  unsigned prof_stride = 6;   // every prof_stride-th launch of the kernel is bracketed
  double prof_flops = 0.0;
  const char* prof_kernel = "gemm (FFN w_1)";  // what the bracketed launches were
  int prof_split = 1;        // hidden slices / K slices of the feed-forward module last run
  int prec = PREC_F32;       // GEMM operand precision (wn_model_set_precision)
  // one host thread per handle: the workspace, the descriptor staging and the
  // current batch are per-handle state.  Entry points take this flag and fail
  // loudly (status -4) instead of corrupting the staging buffer when a second
  // thread enters the same handle (use wn_model_clone for a second thread).
  std::atomic<bool> busy{false};
  // tuning knobs (tune.h): this handle's overrides (wn_model_tune_set; TUNE_INHERIT = follow
  // the process default) and the effective set WN_ENTER resolves for the call in flight
  Tune tune_ovr = tune_all_inherit();
  Tune tune_eff;
  int dbg_layers = -1;       // run only the first n encoder layers
  int dbg_skip_after_norm = 0;
  // fbank tables
  const float* fb_window = nullptr; const float* fb_twiddle = nullptr;
  const float* fb_mel_w = nullptr;
  std::shared_ptr<DevBuf> fb_tab_i = std::make_shared<DevBuf>();
  // context biasing tables (wn_set_context_graph); ctx.keys == nullptr: none
  std::shared_ptr<DevBuf> ctx_buf;
  CtxGraph ctx;
  DevBuf fb_off, fb_nfr;

  int F1() const { return (cfg.feat_dim - 1) / 2; }
  int F2() const { return (F1() - 1) / 2; }
};


// Wait for a stream whose last work item is milliseconds long and whose result the caller needs
// NOW (the searches' result copies): poll instead of sleeping on the completion interrupt, whose
// wake-up costs tens of microseconds per decode; after 50 ms (a stuck or very long queue) fall
// back to the blocking wait.  wn_tune_set("sync_spin", 0) = always the blocking wait (A/B).
inline hipError_t stream_wait(hipStream_t s) {
  if (tune().sync_spin != 0) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0;; ++i) {
      const hipError_t e = hipStreamQuery(s);
      if (e != hipErrorNotReady) return e;
      if ((i & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;
    }
  }
  return hipStreamSynchronize(s);
}

// One host thread per handle at a time; for the duration of the C-ABI call the handle's
// effective tuning set (tune.h: its overrides over the process defaults) is the calling
// thread's tune().
struct HandleGuard {
  wn_model* m;
  bool ok;
  const Tune* saved_tune;
  explicit HandleGuard(wn_model* m_) : m(m_), ok(false), saved_tune(t_tune) {
    bool expected = false;
    ok = m->busy.compare_exchange_strong(expected, true, std::memory_order_acquire);
    if (ok) {
      tune_resolve(m->tune_ovr, &m->tune_eff);
      t_tune = &m->tune_eff;
    }
  }
  ~HandleGuard() {
    if (ok) {
      t_tune = saved_tune;
      m->busy.store(false, std::memory_order_release);
    }
  }
};
#define WN_ENTER(m)                                                              \
  HandleGuard handle_guard(m);                                                   \
  if (!handle_guard.ok) {                                                        \
    ::wn::set_error("this wn_model handle is in use by another host thread; "    \
                    "one thread per handle (wn_model_clone gives a second one)"); \
    return -4;                                                                   \
  }

extern thread_local const std::map<const float*, wn_model::MxW>* t_mx;
// plane images of the current model's weights and its activation-image scratch: linear()
// routes the large fp32 GEMMs to the six-product kernel through them (gemm_x6.hip)
extern thread_local const std::map<const float*, const void*>* t_x6;
extern thread_local DevBuf* t_x6_a;

struct PrecisionScope {
  int saved;
  const float* s_f32; const void* s_bf16; int64_t s_elems;
  const std::map<const float*, wn_model::MxW>* s_mx;
  const std::map<const float*, const void*>* s_x6; DevBuf* s_x6_a;
  explicit PrecisionScope(const wn_model* m)
      : saved(t_gemm_prec), s_f32(t_wslab_f32), s_bf16(t_wslab_bf16),
        s_elems(t_wslab_elems), s_mx(t_mx), s_x6(t_x6), s_x6_a(t_x6_a) {
    t_mx = (m->fp8_ffn && m->mx_at) ? m->mx_at.get() : nullptr;
    t_x6 = m->x6_at ? m->x6_at.get() : nullptr;
    t_x6_a = const_cast<DevBuf*>(&m->x6_lin);
    t_gemm_prec = m->prec;
    const bool img = m->prec == PREC_BF16 && m->weights_bf16 && m->weights_bf16->p;
    t_wslab_f32 = img ? m->weights->as<float>() : nullptr;
    t_wslab_bf16 = img ? m->weights_bf16->p : nullptr;
    t_wslab_elems = img ? m->n_weight_elems : 0;
  }
  ~PrecisionScope() {
    t_gemm_prec = saved;
    t_wslab_f32 = s_f32; t_wslab_bf16 = s_bf16; t_wslab_elems = s_elems;
    t_mx = s_mx;
    t_x6 = s_x6; t_x6_a = s_x6_a;
  }
};

// ---- engine (model.hip) -------------------------------------------------------------------

bool bf16_store_active();
int upload_desc(wn_model* m, DevBuf& buf, const std::vector<int>& v, hipStream_t s);
int linear(const Linear& l, const float* A, int lda, float* C, int ldc, int M, hipStream_t s,
           int act = ACT_NONE, const float* resid = nullptr, int ldr = 0, float alpha = 1.0f,
           bool glu = false, bool a_bf16 = false, bool c_bf16 = false);
int ln(const Norm& n, const float* x, float* y, int M, int D, float eps, hipStream_t s,
       bool y_bf16 = false);
int build_x6_images(wn_model* m);
int vocab_linear(wn_model* m, const Linear& l, const float* A, int lda, float* C, int ldc, int M,
                 hipStream_t s);
int ffn_x6_split(int M, int F);
int ffn_x6_pair(wn_model* m, const Linear& w1, const Linear& w2, int act, const float* A, int M,
                hipStream_t s);
int set_layout(wn_model* m, int B, int Tp, const std::vector<int>& off,
               const std::vector<int>& len, int rows, hipStream_t s);
int subsample_conv2d4(wn_model* m, const float* feats_dev, const int32_t* feat_lens_host, int B,
                      int T, int32_t* enc_lens_host, int pos0, hipStream_t s);
int encode_gate_wait(wn_model* m, hipStream_t s);
int encoder_layers(wn_model* m, int chunk, int left, hipStream_t s);
int encoder_layers_chunk(wn_model* m, int n_sess, int R, const int* offsets,
                         std::vector<ChunkSess>& sess, float* out, hipStream_t s);
int encode_transformer(wn_model* m, const float* feats_dev, const int32_t* feat_lens_host, int B,
                       int T, float* enc_out_dev, int32_t* enc_lens_host, hipStream_t s);

