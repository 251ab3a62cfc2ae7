// Host side of libwenet_amd: weight ingestion / re-layout, workspace arena,
// the encoder / CTC / search / rescoring launch sequences and the C ABI
// (include/wenet_amd.h).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/wenet_amd.h"
#include "kernels.h"

namespace wn {

static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }

namespace {

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
  int end(hipStream_t s) {
    WN_HIP(hipEventRecord(ev, s));
    pending = true;
    return 0;
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
  DevBuf logits, topk_val, topk_idx;
  // searches
  DevBuf pb_dbg;
  DevBuf g_tok, g_len, pb_pool, pb_nh, pb_len, pb_tlen, pb_tok, pb_tim, pb_score;
  // rescoring
  DevBuf r_tok, r_rtok, r_pos, r_tgt, r_rtgt, r_qoff, r_qlen, r_kvoff, r_kvlen;
  DevBuf r_x, r_t1, r_t2, r_qkv, r_h, r_mem, r_logits, r_out;
  DevBuf r_mem_all;            // per-layer cross-attention K/V of the current batch
  DevBuf ab_cache, ab_state;   // `attention` mode: self-attention K|V cache, beam state
  bool mem_cache_valid = false;

  Stager stage;
  // optional HIP-event bracket around the FFN w_1 GEMM launches (the kernel
  // the roofline is quoted on); see wn_profile_*.
  std::vector<hipEvent_t> prof_ev;
  size_t prof_used = 0;
  bool prof_on = false;
  unsigned prof_seq = 0;
  double prof_flops = 0.0;
  const char* prof_kernel = "gemm (FFN w_1)";  // what the bracketed launches were
  int prof_split = 1;        // hidden slices / K slices of the feed-forward module last run
  int prec = PREC_F32;       // GEMM operand precision (wn_model_set_precision)
  // one host thread per handle: the workspace, the descriptor staging and the
  // current batch are per-handle state.  Entry points take this flag and fail
  // loudly (status -4) instead of corrupting the staging buffer when a second
  // thread enters the same handle (use wn_model_clone for a second thread).
  std::atomic<bool> busy{false};
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

namespace {

// Makes the handle's GEMM operand precision current for the calling thread for
// the duration of one C-ABI call (every GEMM launch reads t_gemm_prec).
struct HandleGuard {
  wn_model* m;
  bool ok;
  explicit HandleGuard(wn_model* m_) : m(m_), ok(false) {
    bool expected = false;
    ok = m->busy.compare_exchange_strong(expected, true, std::memory_order_acquire);
  }
  ~HandleGuard() { if (ok) m->busy.store(false, std::memory_order_release); }
};
#define WN_ENTER(m)                                                              \
  HandleGuard handle_guard(m);                                                   \
  if (!handle_guard.ok) {                                                        \
    ::wn::set_error("this wn_model handle is in use by another host thread; "    \
                    "one thread per handle (wn_model_clone gives a second one)"); \
    return -4;                                                                   \
  }

thread_local const std::map<const float*, wn_model::MxW>* t_mx = nullptr;
// plane images of the current model's weights and its activation-image scratch: linear()
// routes the large fp32 GEMMs to the six-product kernel through them (gemm_x6.hip)
thread_local const std::map<const float*, const void*>* t_x6 = nullptr;
thread_local DevBuf* t_x6_a = nullptr;
// fp8 mode: smallest number of 256 x 256 tiles of an FFN GEMM pair for which the MXFP8
// kernels are used (below it the bf16 kernels fill the chip better); tests set 0
int g_fp8_min_tiles = 192;
// bf16-storage form, encoders without the rel-pos term: the QKV GEMM writes bf16 and the
// attention kernel reads it (1, default); 0 keeps fp32 Q / K / V (A/B, tests)
int g_qkv_bf16 = 1;

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

// bf16-storage form of the bf16 mode: LayerNorm output, FFN hidden and attention
// context are written as bf16 (their only consumers are GEMMs that round them to
// bf16 first thing), the GEMMs read the bf16 image of the weight slab.
bool bf16_store_active() {
  return t_gemm_prec == PREC_BF16 && g_bf16_store != 0 && g_attn_bf16 != 0 &&
         t_wslab_bf16 != nullptr;
}

int upload_desc(wn_model* m, DevBuf& buf, const std::vector<int>& v,
                hipStream_t s) {
  return m->stage.put(buf, v.data(), v.size() * sizeof(int), s);
}

// a_bf16 / c_bf16: A / C are bf16 matrices in the same buffers (lda / ldc stay the
// element counts) -- only under bf16_store_active().
// linear(): GEMMs from this many 0.1 GFLOP on go to the six-product kernel through a split pass
int g_x6_linear_min = 60;   // wn_tune_set("x6_linear_min")

int linear(const Linear& l, const float* A, int lda, float* C, int ldc, int M,
           hipStream_t s, int act = ACT_NONE, const float* resid = nullptr,
           int ldr = 0, float alpha = 1.0f, bool glu = false, bool a_bf16 = false,
           bool c_bf16 = false) {
  // Large fp32 GEMMs (the d = 512 encoders' projections, the decoders' GEMMs over B x N x L
  // rows): split A into planes (one pass, 4 B in / 6 B out) and run the six-product kernel
  // -- worth it from ~6 GFLOP on, where the split is a few per cent of the GEMM it halves.
  if (t_gemm_prec == PREC_F32 && g_gemm_x6 != 0 && g_x6_linear != 0 && t_x6 && t_x6_a && !glu &&
      !a_bf16 && !c_bf16 && l.in % 16 == 0 && l.out % 4 == 0 && lda % 4 == 0 && ldc % 4 == 0 &&
      (resid == nullptr || ldr % 4 == 0) && M >= 512 &&
      2.0 * M * (double)l.out * l.in >= 1e8 * g_x6_linear_min) {
    auto it = t_x6->find(l.w);
    if (it != t_x6->end()) {
      WN_TRY(t_x6_a->ensure(x6_bytes(M, l.in)));
      WN_TRY(x6_split(A, M, l.in, lda, t_x6_a->as<char>(), s));
      X6Args x;
      x.A3 = t_x6_a->as<char>(); x.B3 = it->second; x.M = M; x.N = l.out; x.K = l.in;
      x.epi = 0; x.bias = l.b; x.resid = resid; x.ldr = ldr; x.alpha = alpha; x.act = act;
      x.C = C; x.ldc = ldc;
      return gemm_x6(x, s);
    }
  }
  GemmArgs g;
  g.A = A; g.W = l.w; g.bias = l.b; g.C = C; g.resid = resid;
  g.M = M; g.N = l.out; g.K = l.in; g.lda = lda; g.ldc = ldc; g.ldr = ldr;
  g.alpha = alpha; g.act = act; g.glu = glu;
  g.a_bf16 = a_bf16; g.c_bf16 = c_bf16;
  return gemm_f32(g, s);
}

// FFN w_1 GEMM (SiLU epilogue), optionally bracketed by HIP events
int ffn_w1(wn_model* m, const Linear& l, const float* A, float* C, int M,
           hipStream_t s, int act, bool h16) {
  // every hipEventRecord pair costs ~10 us of idle GPU around the launch
  // (measured in the rocprofv3 trace), so only every 6th launch is bracketed:
  // an unbiased sample of the average launch duration (4 per 12-layer pass)
  if (!m->prof_on || (m->prof_seq++ % 6) != 0)
    return linear(l, A, l.in, C, l.out, M, s, act, nullptr, 0, 1.0f, false, h16, h16);
  if (m->prof_used + 2 > m->prof_ev.size()) {
    for (int i = 0; i < 64; ++i) {
      hipEvent_t e;
      WN_HIP(hipEventCreate(&e));
      m->prof_ev.push_back(e);
    }
  }
  WN_HIP(hipEventRecord(m->prof_ev[m->prof_used], s));
  WN_TRY(linear(l, A, l.in, C, l.out, M, s, act, nullptr, 0, 1.0f, false, h16, h16));
  WN_HIP(hipEventRecord(m->prof_ev[m->prof_used + 1], s));
  m->prof_used += 2;
  m->prof_flops += 2.0 * M * (double)l.out * l.in;
  return 0;
}

int ln(const Norm& n, const float* x, float* y, int M, int D, float eps,
       hipStream_t s, bool y_bf16 = false) {
  return layernorm(x, D, n.w, n.b, y, D, M, D, eps, s, y_bf16);
}

// x += alpha * w_2(act(w_1(LN(x)))) -- one feed-forward module
// (positionwise_feed_forward.py:50-58 inside encoder_layer.py:220-228 / :253-261 /
// encoder_layer.py:117-125).  `ln_done`: t1 already holds LN(x) (fused earlier).
// fp8 mode (WN_PREC_FP8) and shapes the pipelined kernel takes: the LayerNorm writes
// MXFP8, w_1 reads it and writes the hidden tensor as MXFP8 again (block scales from
// its epilogue), w_2 reads that and adds into the fp32 residual stream.
int ffn_w1(wn_model* m, const Linear& l, const float* A, float* C, int M, hipStream_t s,
           int act, bool h16);
int ffn_module(wn_model* m, const Norm& nrm, const Linear& w1, const Linear& w2, int act,
               float alpha, bool ln_done, bool h16, hipStream_t s) {
  const wn_config& c = m->cfg;
  const int d = c.d_model, M = m->rows;
  float* x = m->x.as<float>();
  float* t1 = m->t1.as<float>();
  float* hb = m->hbuf.as<float>();
  bool mx = false;
  const wn_model::MxW *q1 = nullptr, *q2 = nullptr;
  if (t_mx && h16 && !ln_done) {
    auto i1 = t_mx->find(w1.w), i2 = t_mx->find(w2.w);
    const int64_t t256 = (int64_t)cdiv(M, 256) * cdiv(std::min(w1.out, w2.out), 256);
    GemmArgs p1, p2;      // the two launches as gemm_mxfp8 will see them: supported shapes only
    p1.M = p2.M = M; p1.N = w1.out; p1.K = d; p1.lda = d; p1.ldc = w1.out;
    p1.fp8 = p2.fp8 = true; p1.c_mx = true;
    p2.N = d; p2.K = w1.out; p2.lda = w1.out; p2.ldc = d; p2.resid = x; p2.ldr = d;
    if (i1 != t_mx->end() && i2 != t_mx->end() && d % 256 == 0 && w1.out % 128 == 0 &&
        t256 >= g_fp8_min_tiles && gemm_bf16p_supported(p1) && gemm_bf16p_supported(p2)) {
      mx = true; q1 = &i1->second; q2 = &i2->second;
    }
  }
  if (!mx) {
    if (!ln_done) WN_TRY(ln(nrm, x, t1, M, d, c.norm_eps, s, h16));
    WN_TRY(ffn_w1(m, w1, t1, hb, M, s, act, h16));
    return linear(w2, hb, w1.out, x, d, M, s, ACT_NONE, x, d, alpha, false, h16);
  }
  const int pitch = cdiv(M, 256) * 256;
  WN_TRY(m->mx_sa.ensure((size_t)(d / 128) * pitch * 4));
  WN_TRY(m->mx_sh.ensure((size_t)(w1.out / 128) * pitch * 4));
  WN_TRY(layernorm_mx(x, d, nrm.w, nrm.b, t1, m->mx_sa.as<unsigned>(), pitch, M, d,
                      c.norm_eps, s));
  GemmArgs g;
  g.A = t1; g.bias = w1.b; g.C = hb; g.M = M; g.N = w1.out; g.K = d;
  g.lda = d; g.ldc = w1.out; g.act = act; g.fp8 = true; g.c_mx = true;
  g.a_scale = m->mx_sa.as<unsigned>(); g.a_scale_pitch = pitch;
  g.w_scale = q1->scale; g.w_scale_pitch = w1.out;
  g.c_scale = m->mx_sh.as<unsigned>(); g.c_scale_pitch = pitch;
  const bool bracket = m->prof_on && (m->prof_seq++ % 6) == 0;
  if (bracket) {
    if (m->prof_used + 2 > m->prof_ev.size())
      for (int i = 0; i < 64; ++i) {
        hipEvent_t e;
        WN_HIP(hipEventCreate(&e));
        m->prof_ev.push_back(e);
      }
    WN_HIP(hipEventRecord(m->prof_ev[m->prof_used], s));
  }
  WN_TRY(gemm_mxfp8(g, q1->q, s));
  if (bracket) {
    WN_HIP(hipEventRecord(m->prof_ev[m->prof_used + 1], s));
    m->prof_used += 2;
    m->prof_flops += 2.0 * M * (double)w1.out * d;
  }
  GemmArgs h;
  h.A = hb; h.bias = w2.b; h.C = x; h.resid = x; h.M = M; h.N = d; h.K = w1.out;
  h.lda = w1.out; h.ldc = d; h.ldr = d; h.alpha = alpha; h.fp8 = true;
  h.a_scale = m->mx_sh.as<unsigned>(); h.a_scale_pitch = pitch;
  h.w_scale = q2->scale; h.w_scale_pitch = d;
  return gemm_mxfp8(h, q2->q, s);
}

// Plane images (gemm_x6.hip) of the encoder's feed-forward weights, once per model.
int build_x6_images(wn_model* m) {
  std::vector<const Linear*> ws;
  for (const auto& L : m->layers) { ws.push_back(&L.ffm1); ws.push_back(&L.ffm2);
                                    ws.push_back(&L.ff1); ws.push_back(&L.ff2);
                                    ws.push_back(&L.qkv); ws.push_back(&L.out);
                                    ws.push_back(&L.pw2); }
  for (const Decoder* D : {&m->left, &m->right})
    for (const auto& L : D->layers) {
      ws.push_back(&L.self_qkv); ws.push_back(&L.self_out); ws.push_back(&L.src_q);
      ws.push_back(&L.src_kv); ws.push_back(&L.src_out); ws.push_back(&L.ff1);
      ws.push_back(&L.ff2);
    }
  // (the Transformer encoder of the Whisper configuration runs its GEMMs on v_mfma_f32 or,
  // in the bf16 / fp8 modes, on the low-precision kernels: no images for tf_layers)
  if (m->conv2.w) ws.push_back(&m->conv2);   // [d][(ky*3+kx)*d + c]: 16-channel k blocks per tap
  std::vector<const Linear*> vocab;          // V rows; the image pads them to a multiple of 32
  if (m->ctc.w) vocab.push_back(&m->ctc);
  for (const Decoder* D : {&m->left, &m->right})
    if (D->out.w) vocab.push_back(&D->out);
  for (const Linear* l : vocab) ws.push_back(l);
  size_t bytes = 0;
  for (const Linear* l : ws)
    if (l->w && l->in % 16 == 0) bytes += x6_bytes(l->out, l->in);
  auto buf = std::make_shared<DevBuf>();
  auto at = std::make_shared<std::map<const float*, const void*>>();
  if (bytes > 0) {
    WN_TRY(buf->ensure(bytes));
    char* p = buf->as<char>();
    for (const Linear* l : ws) {
      if (!l->w || l->in % 16 != 0 || at->count(l->w)) continue;
      WN_TRY(x6_split(l->w, l->out, l->in, l->in, p, nullptr));
      (*at)[l->w] = p;
      p += x6_bytes(l->out, l->in);
    }
  }
  m->weights_x6 = buf;
  m->x6_at = at;
  // fused six-product feed-forward module (ffn_x6f.hip, d_model 256): the second layer's image
  // with the k slots of a 16-unit block in the order a lane holds its hidden values
  auto pbuf = std::make_shared<DevBuf>();
  auto pat = std::make_shared<std::map<const float*, const void*>>();
  if (m->cfg.d_model == 256) {
    std::vector<const Linear*> w2s;
    for (const auto& L : m->layers) { w2s.push_back(&L.ffm2); w2s.push_back(&L.ff2); }
    size_t pbytes = 0;
    for (const Linear* l : w2s)
      if (l->w && l->in % 64 == 0 && l->out == 256) pbytes += x6_bytes(l->out, l->in);
    if (pbytes > 0) {
      WN_TRY(pbuf->ensure(pbytes));
      char* q = pbuf->as<char>();
      for (const Linear* l : w2s) {
        if (!l->w || l->in % 64 != 0 || l->out != 256 || pat->count(l->w)) continue;
        WN_TRY(x6_split_perm(l->w, l->out, l->in, l->in, q, nullptr));
        (*pat)[l->w] = q;
        q += x6_bytes(l->out, l->in);
      }
    }
  }
  m->weights_x6p = pbuf;
  m->x6p_at = pat;
  // the six-product kernel stores 16-B pieces: the vocabulary-sized layers run with N = V
  // rounded up to 4 (the image rows past V are zero, their bias too) and their logits rows
  // get that pitch
  auto b4 = std::make_shared<DevBuf>();
  auto bmap = std::make_shared<std::map<const float*, const float*>>();
  size_t b4_floats = 0;
  for (const Linear* l : vocab)
    if (l->b && l->out % 4 != 0) b4_floats += (size_t)(l->out + 3) / 4 * 4;
  if (b4_floats > 0) {
    WN_TRY(b4->ensure(b4_floats * sizeof(float)));
    WN_HIP(hipMemsetAsync(b4->p, 0, b4_floats * sizeof(float), nullptr));
    float* q = b4->as<float>();
    for (const Linear* l : vocab) {
      if (!l->b || l->out % 4 == 0 || bmap->count(l->w)) continue;
      WN_HIP(hipMemcpyAsync(q, l->b, (size_t)l->out * sizeof(float), hipMemcpyDeviceToDevice,
                            nullptr));
      (*bmap)[l->w] = q;
      q += (l->out + 3) / 4 * 4;
    }
  }
  m->bias4_buf = b4;
  m->bias4 = bmap;
  return 0;
}

// A vocabulary-sized layer (N = V, any V) into a logits buffer whose rows have a pitch of V
// rounded up to 4: the six-product GEMM when the layer has a plane image (and, for V % 4 !=
// 0, a padded bias), else linear().
int vocab_linear(wn_model* m, const Linear& l, const float* A, int lda, float* C, int M,
                 hipStream_t s) {
  const int V = l.out, V4 = (V + 3) / 4 * 4;
  const void* w6 = nullptr;
  const float* bias = l.b;
  if (t_gemm_prec == PREC_F32 && g_gemm_x6 != 0 && g_x6_linear != 0 && m->x6_at &&
      l.in % 16 == 0 && lda % 4 == 0 && M >= 512) {
    auto it = m->x6_at->find(l.w);
    if (it != m->x6_at->end()) w6 = it->second;
    if (w6 && V != V4 && l.b) {     // ragged V: the padded copy of the bias, or no x6
      bias = nullptr;
      if (m->bias4) {
        auto ib = m->bias4->find(l.w);
        if (ib != m->bias4->end()) bias = ib->second;
      }
      if (!bias) w6 = nullptr;
    }
  }
  if (!w6) return linear(l, A, lda, C, V4, M, s);
  WN_TRY(m->x6_lin.ensure(x6_bytes(M, l.in)));
  WN_TRY(x6_split(A, M, l.in, lda, m->x6_lin.as<char>(), s));
  X6Args x;
  x.A3 = m->x6_lin.as<char>(); x.B3 = w6; x.M = M; x.N = V4; x.K = l.in;
  x.epi = 0; x.bias = bias; x.C = C; x.ldc = V4;
  return gemm_x6(x, s);
}

// hidden split of the x6 FFN's second GEMM: K slices so that 128-row tiles x slices fill
// the CUs once
int ffn_x6_split(int M, int F) {
  if (g_x6_ffn_s > 0 && (F / 16) % g_x6_ffn_s == 0) return g_x6_ffn_s;   // A/B knob
  int S = 1;
  while (S < 16 && cdiv(M, 128) * (S * 2) <= 256 && (F / 16) % (S * 2) == 0) S *= 2;
  return S;
}

// fp32 feed-forward module on the bf16 matrix cores (gemm_x6.hip): t1 = LN(x) is in place;
// split it into planes, w_1 + activation straight into the plane image of the hidden
// tensor, w_2 as K-slice partials in m->ffn_part.  Returns the slice count (0: not taken).
int ffn_x6_try(wn_model* m, const Linear& w1, const Linear& w2, int act, hipStream_t s) {
  const int d = m->cfg.d_model, M = m->rows, F = w1.out;
  // (d: the widths ffn_reduce_ln takes)
  if (t_gemm_prec != PREC_F32 || g_gemm_x6 == 0 || !m->x6_at || w1.out != w2.in ||
      !(d == 256 || d == 512) || F % 16 != 0 || (M < 512 && g_gemm_x6 != 2))
    return 0;
  auto i1 = m->x6_at->find(w1.w), i2 = m->x6_at->find(w2.w);
  if (i1 == m->x6_at->end() || i2 == m->x6_at->end()) return 0;
  static thread_local int tick = 0;
  if (g_ffn_x6f != 0 && g_x6_af32 == 0 && m->x6p_at && ffn_x6f_supported(M, d, F, act)) {
    // hidden tensor on chip (ffn_x6f.hip)
    auto ip = m->x6p_at->find(w2.w);
    if (ip != m->x6p_at->end()) {
      FfnX6Args a;
      a.S = ffn_x6f_split(M, F);
      if (m->ffn_part.ensure((size_t)a.S * M * d * sizeof(float)) != 0) return -1;
      a.X = m->t1.as<float>(); a.ldx = d; a.W13 = i1->second; a.W2p = ip->second; a.b1 = w1.b;
      a.P = m->ffn_part.as<float>(); a.M = M; a.D = d; a.F = F; a.act = act;
      const bool br = m->prof_on && (tick++ % 6) == 0;
      if (br) {
        if (m->prof_used + 2 > m->prof_ev.size())
          for (int i = 0; i < 64; ++i) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return -1;
            m->prof_ev.push_back(e);
          }
        (void)hipEventRecord(m->prof_ev[m->prof_used], s);
      }
      if (ffn_x6f(a, s) != 0) return -1;
      if (br) {
        (void)hipEventRecord(m->prof_ev[m->prof_used + 1], s);
        m->prof_used += 2;
        m->prof_flops += 4.0 * M * (double)F * d;     // both contractions (x 6 MFMA products)
        m->prof_kernel = "ffn_x6f_kernel (FFN w_1 + act + w_2, six bf16 plane products)";
      }
      m->prof_split = a.S;
      return a.S;
    }
  }
  const int S = ffn_x6_split(M, F);
  if (m->ffn_part.ensure((size_t)S * M * d * sizeof(float)) != 0) return -1;
  // plane images (x6_split of t1, w_1 writes the hidden planes); g_x6_af32 (A/B knob): the A
  // operands stay plain fp32 (t1, the hidden tensor in hbuf) and are split in registers
  const bool af32 = g_x6_af32 != 0 && (int64_t)M * F * 4 < ((int64_t)1 << 31);
  X6Args g1;
  g1.B3 = i1->second; g1.M = M; g1.N = F; g1.K = d; g1.bias = w1.b; g1.act = act;
  if (af32) {
    if (m->hbuf.ensure((size_t)M * F * sizeof(float)) != 0) return -1;
    g1.A = m->t1.as<float>(); g1.lda = d; g1.a_bytes = (int64_t)M * d * 4;
    g1.epi = 0; g1.C = m->hbuf.as<float>(); g1.ldc = F;
  } else {
    if (m->x6_a.ensure(x6_bytes(M, d)) != 0 || m->x6_h.ensure(x6_bytes(M, F)) != 0) return -1;
    if (x6_split(m->t1.as<float>(), M, d, d, m->x6_a.as<char>(), s) != 0) return -1;
    g1.A3 = m->x6_a.as<char>(); g1.epi = 2; g1.C3 = m->x6_h.as<char>();
  }
  if (g_x6_nw4 & 1) g1.bm = 256;                      // A/B: the 256-row tiles for w_1
  if (g_x6_nw4 & 4) g1.prio_split = cdiv(M, 128) * cdiv(F, 256) / 2;
  const bool bracket = m->prof_on && (tick++ % 6) == 0;
  if (bracket) {
    if (m->prof_used + 2 > m->prof_ev.size())
      for (int i = 0; i < 64; ++i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return -1;
        m->prof_ev.push_back(e);
      }
    (void)hipEventRecord(m->prof_ev[m->prof_used], s);
  }
  if (gemm_x6(g1, s) != 0) return -1;
  if (bracket) {
    (void)hipEventRecord(m->prof_ev[m->prof_used + 1], s);
    m->prof_used += 2;
    m->prof_flops += 2.0 * M * (double)F * d;       // the contraction (x 6 MFMA products)
    m->prof_kernel = "gemm_x6_kernel (FFN w_1 + act, six bf16 plane products)";
  }
  X6Args g2;
  g2.B3 = i2->second; g2.M = M; g2.N = d; g2.K = F;
  g2.epi = 1; g2.ksplit = S; g2.C = m->ffn_part.as<float>();
  if (af32) { g2.A = m->hbuf.as<float>(); g2.lda = F; g2.a_bytes = (int64_t)M * F * 4; }
  else g2.A3 = m->x6_h.as<char>();
  if (g_x6_nw4 & 2) g2.nw = 8;                        // A/B: the 8-wave form of the 128-row tile
  if (gemm_x6(g2, s) != 0) return -1;
  m->prof_split = S;
  return S;
}

// fp32 fused feed-forward module (ffn_fused.hip): t1 = LN(x) is in place; leaves the
// hidden-slice partials in m->ffn_part and returns S (0: shape not taken, caller runs
// the two-GEMM path).  Every 6th launch is bracketed for the roofline (wn_profile_*).
int ffn_fused_try(wn_model* m, const Linear& w1, const Linear& w2, int act, hipStream_t s) {
  const wn_config& c = m->cfg;
  const int d = c.d_model, M = m->rows;
  if (const int s6 = ffn_x6_try(m, w1, w2, act, s)) return s6;
  if (t_gemm_prec != PREC_F32 || g_ffn_fused == 0 || w1.out != w2.in ||
      !ffn_fused_supported(M, d, w1.out, act))
    return 0;
  FfnArgs a;
  a.X = m->t1.as<float>(); a.W1 = w1.w; a.b1 = w1.b; a.W2 = w2.w;
  a.M = M; a.D = d; a.F = w1.out; a.S = ffn_fused_split(M, d, w1.out); a.act = act;
  if (m->ffn_part.ensure((size_t)a.S * M * d * sizeof(float)) != 0) return -1;
  a.P = m->ffn_part.as<float>();
  const bool bracket = m->prof_on && (m->prof_seq++ % 6) == 0;
  if (bracket) {
    if (m->prof_used + 2 > m->prof_ev.size())
      for (int i = 0; i < 64; ++i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return -1;
        m->prof_ev.push_back(e);
      }
    (void)hipEventRecord(m->prof_ev[m->prof_used], s);
  }
  if (ffn_fused(a, s) != 0) return -1;
  if (bracket) {
    (void)hipEventRecord(m->prof_ev[m->prof_used + 1], s);
    m->prof_used += 2;
    m->prof_flops += 4.0 * M * (double)w1.out * d;    // both contractions
    m->prof_kernel = "ffn_fused_kernel (FFN w_1 + act + w_2)";
  }
  m->prof_split = a.S;
  return a.S;
}

// ---- set the per-utterance row layout of the current batch -----------------
int set_layout(wn_model* m, int B, int Tp, const std::vector<int>& off,
               const std::vector<int>& len, int rows, hipStream_t s) {
  m->B = B; m->Tp = Tp; m->off = off; m->len = len; m->rows = rows;
  m->mem_cache_valid = false;
  std::vector<int> row_utt(std::max(rows, 1), -1);
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < len[b]; ++t) row_utt[off[b] + t] = b;
  WN_TRY(m->stage.begin((size_t)(rows + 4 * B + 64) * sizeof(int) + 1024));
  WN_TRY(upload_desc(m, m->d_off, off, s));
  WN_TRY(upload_desc(m, m->d_len, len, s));
  WN_TRY(upload_desc(m, m->d_row_utt, row_utt, s));
  m->ctc_valid = false;
  return 0;
}

// GlobalCMVN + Conv2dSubsampling4 + RelPositionalEncoding scale for a padded
// (B, T, F) feature batch: sets the row layout and leaves x = embed(xs) in m->x
// (encoder.py:155-157, subsampling.py:203-228, embedding.py:134-147).  `pos0` is
// the position of the first output frame (streaming offset).
int subsample_conv2d4(wn_model* m, const float* feats_dev,
                      const int32_t* feat_lens_host, int B, int T,
                      int32_t* enc_lens_host, int pos0, hipStream_t s) {
  const wn_config& c = m->cfg;
  const int d = c.d_model, F1 = m->F1(), F2 = m->F2();
  const int Tp = ((T - 1) / 2 - 1) / 2;
  WN_CHECK(pos0 + Tp <= c.max_pos, "utterance longer than the positional table");
  std::vector<int> off2(B), len2(B), off1(B), len1(B);
  int M = 0, M1 = 0, max_t1 = 0;
  for (int b = 0; b < B; ++b) {
    const int L = feat_lens_host[b];
    WN_CHECK(L >= 0 && L <= T, "wn_encode: feature length out of range");
    // mask[:, :, 2::2][:, :, 2::2] (subsampling.py:228): frames 6 + 4k < L
    const int l2 = L > 6 ? (L - 7) / 4 + 1 : 0;
    off2[b] = M; len2[b] = l2; M += l2;
    const int l1 = l2 > 0 ? 2 * l2 + 1 : 0;
    off1[b] = M1; len1[b] = l1; M1 += l1;
    max_t1 = std::max(max_t1, l1);
    if (enc_lens_host) enc_lens_host[b] = l2;
  }
  WN_TRY(set_layout(m, B, Tp, off2, len2, M, s));
  if (M == 0) WN_TRY(m->stage.end(s));
  if (M > 0) {
    WN_TRY(upload_desc(m, m->d_off1, off1, s));
    WN_TRY(upload_desc(m, m->d_len1, len1, s));
    WN_TRY(m->stage.end(s));
    WN_TRY(m->c1.ensure((size_t)M1 * F1 * d * sizeof(float)));
    WN_TRY(m->c2.ensure((size_t)M * F2 * d * sizeof(float)));
    WN_TRY(m->x.ensure((size_t)M * d * sizeof(float)));
    WN_TRY(m->t1.ensure((size_t)M * d * sizeof(float)));
    WN_TRY(m->t2.ensure((size_t)M * d * sizeof(float)));
    WN_TRY(m->hbuf.ensure((size_t)M * c.ffn_dim * sizeof(float)));
    WN_TRY(m->qkv.ensure((size_t)M * 3 * d * sizeof(float)));
    WN_TRY(m->d_a_row_off.ensure((size_t)M * F2 * sizeof(int64_t)));
    // fp32 on the bf16 matrix cores (gemm_x6.hip): conv1 writes the plane image of its
    // output, conv2 gathers its rows from it
    const void* w6 = nullptr;
    if (t_gemm_prec == PREC_F32 && g_gemm_x6 != 0 && g_x6_conv != 0 && m->x6_at && d % 32 == 0 &&
        F1 <= 64 &&
        (M * F2 >= 4096 || g_gemm_x6 == 2)) {
      auto it = m->x6_at->find(m->conv2.w);
      if (it != m->x6_at->end()) w6 = it->second;
    }
    if (w6 && g_x6_af32 != 0 && (int64_t)M1 * F1 * d * 4 < ((int64_t)1 << 31)) {
      // conv1 as always (fp32, channels last); conv2 gathers its A rows from it, 64 B per
      // pixel and k block, and splits them in registers
      WN_TRY(m->c1.ensure((size_t)M1 * F1 * d * sizeof(float)));
      Conv1Args c1;
      c1.feats = feats_dev; c1.mean = m->cmvn_mean; c1.istd = m->cmvn_istd;
      c1.w = m->conv1_w; c1.bias = m->conv1_b; c1.out = m->c1.as<float>();
      c1.t1_off = m->d_off1.as<int>(); c1.t1_len = m->d_len1.as<int>();
      c1.B = B; c1.T = T; c1.F = c.feat_dim; c1.F1 = F1; c1.C = d; c1.max_t1 = max_t1;
      WN_TRY(cmvn_conv1_relu(c1, s));
      int* pix = reinterpret_cast<int*>(m->d_a_row_off.as<int64_t>());
      hipLaunchKernelGGL(build_conv2_pix_kernel, dim3(cdiv(M * F2, 256)), dim3(256), 0, s,
                         m->d_row_utt.as<int>(), m->d_off.as<int>(), m->d_off1.as<int>(), M,
                         F1, F2, 2, pix);
      WN_HIP(hipGetLastError());
      X6Args g;
      g.A = m->c1.as<float>(); g.a_bytes = (int64_t)M1 * F1 * d * 4;
      g.B3 = w6; g.M = M * F2; g.N = d; g.K = 9 * d;
      g.epi = 0; g.bias = m->conv2.b; g.act = ACT_RELU; g.C = m->c2.as<float>(); g.ldc = d;
      g.a_pix = pix; g.conv_kbc = d / 16; g.bm = g_x6_conv_bm;
      for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) g.tap_delta[ky * 3 + kx] = ky * F1 + kx;
      WN_TRY(gemm_x6(g, s));
      WN_TRY(linear(m->sub_out, m->c2.as<float>(), F2 * d, m->x.as<float>(), d, M,
                    s, ACT_NONE, nullptr, 0, sqrtf((float)d)));
      return 0;
    }
    if (w6) {
      const int tiles = cdiv(M1 * F1, 32);
      WN_TRY(m->c1.ensure(x6_bytes(M1 * F1, d)));
      Conv1Args c1;
      c1.feats = feats_dev; c1.mean = m->cmvn_mean; c1.istd = m->cmvn_istd;
      c1.w = m->conv1_w; c1.bias = m->conv1_b; c1.out = nullptr;
      c1.out3 = m->c1.as<char>(); c1.tiles = tiles;
      c1.t1_off = m->d_off1.as<int>(); c1.t1_len = m->d_len1.as<int>();
      c1.B = B; c1.T = T; c1.F = c.feat_dim; c1.F1 = F1; c1.C = d; c1.max_t1 = max_t1;
      WN_TRY(cmvn_conv1_relu(c1, s));
      int* pix = reinterpret_cast<int*>(m->d_a_row_off.as<int64_t>());
      hipLaunchKernelGGL(build_conv2_pix_kernel, dim3(cdiv(M * F2, 256)), dim3(256), 0, s,
                         m->d_row_utt.as<int>(), m->d_off.as<int>(), m->d_off1.as<int>(), M,
                         F1, F2, 1, pix);
      WN_HIP(hipGetLastError());
      X6Args g;
      g.A3 = m->c1.as<char>(); g.B3 = w6; g.M = M * F2; g.N = d; g.K = 9 * d;
      g.epi = 0; g.bias = m->conv2.b; g.act = ACT_RELU; g.C = m->c2.as<float>(); g.ldc = d;
      g.a_pix = pix; g.a_tiles = tiles; g.conv_kbc = d / 16; g.bm = g_x6_conv_bm;
      g.conv_taps = g_x6_conv_order ? 9 : 0;
      const int ne = (F1 + 1) / 2;
      for (int ky = 0; ky < 3; ++ky) {
        g.tap_delta[ky * 3 + 0] = ky * F1;            // f1 = 2 f2     (even, position f2)
        g.tap_delta[ky * 3 + 1] = ky * F1 + ne;       // f1 = 2 f2 + 1 (odd, position ne + f2)
        g.tap_delta[ky * 3 + 2] = ky * F1 + 1;        // f1 = 2 f2 + 2 (even, position f2 + 1)
      }
      WN_TRY(gemm_x6(g, s));
      WN_TRY(linear(m->sub_out, m->c2.as<float>(), F2 * d, m->x.as<float>(), d, M,
                    s, ACT_NONE, nullptr, 0, sqrtf((float)d)));
      return 0;
    }
    // GlobalCMVN + conv1 + ReLU                        encoder.py:155, subsampling.py:188
    Conv1Args c1;
    c1.feats = feats_dev; c1.mean = m->cmvn_mean; c1.istd = m->cmvn_istd;
    c1.w = m->conv1_w; c1.bias = m->conv1_b; c1.out = m->c1.as<float>();
    c1.t1_off = m->d_off1.as<int>(); c1.t1_len = m->d_len1.as<int>();
    c1.B = B; c1.T = T; c1.F = c.feat_dim; c1.F1 = F1; c1.C = d; c1.max_t1 = max_t1;
    WN_TRY(cmvn_conv1_relu(c1, s));
    // conv2 + ReLU as an implicit GEMM                 subsampling.py:191-192
    hipLaunchKernelGGL(build_conv2_rows_kernel, dim3(cdiv(M * F2, 256)),
                       dim3(256), 0, s, m->d_row_utt.as<int>(),
                       m->d_off.as<int>(), m->d_off1.as<int>(), M, F1, F2, d,
                       m->d_a_row_off.as<int64_t>());
    WN_HIP(hipGetLastError());
    GemmArgs g;
    g.A = m->c1.as<float>(); g.W = m->conv2.w; g.bias = m->conv2.b;
    g.C = m->c2.as<float>(); g.M = M * F2; g.N = d; g.K = 9 * d; g.ldc = d;
    g.act = ACT_RELU; g.a_row_off = m->d_a_row_off.as<int64_t>();
    g.conv_C = d; g.conv_sy = (int64_t)F1 * d; g.conv_sx = d;
    WN_TRY(gemm_f32(g, s));
    // Linear(d*F2 -> d) * sqrt(d)                      subsampling.py:225-226, embedding.py:144
    WN_TRY(linear(m->sub_out, m->c2.as<float>(), F2 * d, m->x.as<float>(), d, M,
                  s, ACT_NONE, nullptr, 0, sqrtf((float)d)));
  }
  return 0;
}

int encoder_layers(wn_model* m, int chunk, int left, hipStream_t s) {
  const wn_config& c = m->cfg;
  const int d = c.d_model, M = m->rows;
  float* x = m->x.as<float>();
  float* t1 = m->t1.as<float>();
  float* t2 = m->t2.as<float>();
  float* hb = m->hbuf.as<float>();
  float* qkv = m->qkv.as<float>();
  const float eps = c.norm_eps;
  int max_len = 0;
  for (int b = 0; b < m->B; ++b) max_len = std::max(max_len, m->len[b]);
  // chunk mask (mask.py:126-198, decode-time branches)
  int mask_mode = 0, cs = 0, lc = -1;
  if (c.use_dynamic_chunk) {
    if (chunk > 0) { mask_mode = 2; cs = chunk; lc = left; }
  } else if (c.static_chunk_size > 0) {
    mask_mode = 2; cs = c.static_chunk_size; lc = left;
  }
  const int n_run = m->dbg_layers >= 0 ? std::min(m->dbg_layers, c.n_layers)
                                      : c.n_layers;
  // bf16-storage mode: the LayerNorm outputs (t1), the FFN hidden (hb) and the
  // attention context (t2) hold bf16; the GLU output / depthwise-conv tensors
  // stay fp32 (the depthwise kernel is fp32)
  const bool h16 = bf16_store_active();
  for (int li = 0; li < n_run; ++li) {
    const EncLayer& L = m->layers[li];
    // x += 0.5 * FFN_macaron(LN(x))                 encoder_layer.py:220-228
    // (for li > 0 the previous layer's tail already left LN(x) in t1)
    // fp32: fused FFN (hidden tensor stays on chip), its partial reduction carries the
    // residual add and the NEXT LayerNorm (norm_mha)
    int fS = 0;
    if (!h16 && t_gemm_prec == PREC_F32) {
      if (li == 0) WN_TRY(ln(L.norm_ff_mac, x, t1, M, d, eps, s));
      fS = ffn_fused_try(m, L.ffm1, L.ffm2, ACT_SILU, s);
      if (fS < 0) return -2;
    }
    if (fS > 0) {
      WN_TRY(ffn_reduce_ln(x, m->ffn_part.as<float>(), fS, L.ffm2.b, 0.5f, L.norm_mha.w,
                           L.norm_mha.b, nullptr, nullptr, t1, M, d, eps, 0, s));
    } else {
      // (fp8 mode: every feed-forward module normalises for itself -- layernorm_mx writes the
      // MXFP8 operand -- so that ALL of them take the same path; the bf16 / fp32 modes get
      // LN(x) from the previous layer's fused tail)
      WN_TRY(ffn_module(m, L.norm_ff_mac, L.ffm1, L.ffm2, ACT_SILU, 0.5f,
                        (li > 0 && !(t_mx && h16)) || (!h16 && t_gemm_prec == PREC_F32), h16, s));
      // x += MHA(LN(x))                               encoder_layer.py:230-238
      WN_TRY(ln(L.norm_mha, x, t1, M, d, eps, s, h16));
    }
    bool qkv_done = false;
    if (!h16 && t_gemm_prec == PREC_F32 && g_x6r != 0 && g_gemm_x6 != 0 && t_x6 && M >= 512 &&
        gemm_x6r_supported(M, 3 * d, L.qkv.in, 0)) {
      auto it = t_x6->find(L.qkv.w);
      if (it != t_x6->end()) {
        X6RArgs g;
        g.A = t1; g.lda = d; g.W3 = it->second; g.bias = L.qkv.b; g.M = M; g.N = 3 * d;
        g.epi = 0; g.C = qkv; g.ldc = 3 * d;
        WN_TRY(gemm_x6r(g, s));
        qkv_done = true;
      }
    }
    if (!qkv_done)
      WN_TRY(linear(L.qkv, t1, d, qkv, 3 * d, M, s, ACT_NONE, nullptr, 0, 1.0f, false,
                    h16));
    AttnArgs a;
    a.Q = qkv; a.K = qkv + d; a.V = qkv + 2 * d;
    a.ldq = a.ldk = a.ldv = 3 * d;
    a.P = L.pos_tab; a.ldp = d; a.bias_u = L.bias_u; a.bias_v = L.bias_v;
    if (!h16 && t_gemm_prec == PREC_F32 && g_attn_fold == 2 && (d == 256 || d == 512) &&
        d == c.n_heads * 64) {
      // A/B form: the folding as a separate pass (k <- k + p in place, scalars in HBM)
      WN_TRY(m->attn_kbias.ensure((size_t)M * c.n_heads * sizeof(float)));
      WN_TRY(relpos_fold(qkv + d, 3 * d, L.pos_tab, d, L.bias_u, L.bias_v,
                         m->d_row_utt.as<int>(), m->d_off.as<int>(), nullptr,
                         m->attn_kbias.as<float>(), c.n_heads, M, d, s));
      a.kbias = m->attn_kbias.as<float>();
      a.P = nullptr; a.bias_u = a.bias_v = nullptr;
    } else if (!h16 && t_gemm_prec == PREC_F32 && g_attn_fold != 0) {
      // rel-pos folded into the keys as the attention kernel stages them: ONE score
      // contraction (encoder_kernels.hip, attention_kernel FOLD / relpos_fold_kernel)
      a.fold = true;
    }
    a.O = t2; a.ldo = d; a.o_bf16 = h16;
    a.q_off = a.kv_off = m->d_off.as<int>();
    a.q_len = a.kv_len = m->d_len.as<int>();
    a.n_seq = m->B; a.n_heads = c.n_heads; a.max_q_len = max_len;
    a.mask_mode = mask_mode; a.chunk_size = cs; a.left_chunks = lc;
    a.scale = 1.0f / sqrtf(64.0f);
    WN_TRY(attention(a, s));
    // x += out_proj(context); t1 = LN_conv(x)       encoder_layer.py:236-240
    const bool rowln = !h16 && t_gemm_prec == PREC_F32 && gemm_rowln_supported(M, d, d);
    // the same fusion as six bf16 plane products with the A rows in registers (gemm_x6r.hip)
    auto x6r_rowln = [&](const Linear& l, const float* A, const Norm& nrm) -> int {
      if (!(rowln && g_x6r != 0 && g_gemm_x6 != 0 && t_x6 && M >= 512 &&
            gemm_x6r_supported(M, d, l.in, 1)))
        return 1;
      auto it = t_x6->find(l.w);
      if (it == t_x6->end()) return 1;
      X6RArgs g;
      g.A = A; g.lda = d; g.W3 = it->second; g.bias = l.b; g.M = M; g.N = d; g.epi = 1;
      g.resid = x; g.ldr = d; g.alpha = 1.0f; g.x_out = x; g.ldx = d;
      g.ln_w = nrm.w; g.ln_b = nrm.b; g.eps = eps; g.y = t1; g.ldy = d;
      return gemm_x6r(g, s) == 0 ? 0 : -1;
    };
    int xr = x6r_rowln(L.out, t2, L.norm_conv);
    if (xr < 0) return -2;
    if (xr == 0) {
    } else if (rowln) {
      RowLnArgs g;
      g.A = t2; g.lda = d; g.W = L.out.w; g.bias = L.out.b; g.resid = x; g.ldr = d;
      g.alpha = 1.0f; g.x_out = x; g.ldx = d; g.ln_w = L.norm_conv.w; g.ln_b = L.norm_conv.b;
      g.eps = eps; g.y = t1; g.ldy = d; g.M = M; g.N = d; g.K = d;
      WN_TRY(gemm_rowln(g, s));
    } else {
      WN_TRY(linear(L.out, t2, d, x, d, M, s, ACT_NONE, x, d, 1.0f, false, h16));
      // x += Conv(LN(x))                              encoder_layer.py:240-251
      WN_TRY(ln(L.norm_conv, x, t1, M, d, eps, s, h16));
    }
    WN_TRY(linear(L.pw1, t1, d, t2, d, M, s, ACT_NONE, nullptr, 0, 1.0f, true, h16));
    DwConvArgs dw;
    dw.x = t2; dw.ldx = d; dw.wt = L.dw_wt; dw.bias = L.dw_b; dw.cpad = L.cpad;
    dw.ln_w = L.conv_norm.w; dw.ln_b = L.conv_norm.b; dw.norm_mode = c.cnn_norm;
    dw.y = t1; dw.ldy = d;
    dw.row_utt = m->d_row_utt.as<int>(); dw.off = m->d_off.as<int>();
    dw.len = m->d_len.as<int>();
    dw.M = M; dw.D = d; dw.K = c.cnn_kernel; dw.causal = c.causal;
    dw.t_max = m->Tp; dw.eps = 1e-5f;
    WN_TRY(dwconv_ln_silu(dw, s));
    // x += pointwise_conv2(.); t1 = LN_ff(x)        encoder_layer.py:251-255
    xr = x6r_rowln(L.pw2, t1, L.norm_ff);
    if (xr < 0) return -2;
    if (xr == 0) {
    } else if (rowln) {
      RowLnArgs g;
      g.A = t1; g.lda = d; g.W = L.pw2.w; g.bias = L.pw2.b; g.resid = x; g.ldr = d;
      g.alpha = 1.0f; g.x_out = x; g.ldx = d; g.ln_w = L.norm_ff.w; g.ln_b = L.norm_ff.b;
      g.eps = eps; g.y = t1; g.ldy = d; g.M = M; g.N = d; g.K = d;
      WN_TRY(gemm_rowln(g, s));
    } else {
      WN_TRY(linear(L.pw2, t1, d, x, d, M, s, ACT_NONE, x, d));
    }
    // x += 0.5 * FFN(LN(x)); x = LN(x)              encoder_layer.py:253-263
    fS = 0;
    if (!h16 && t_gemm_prec == PREC_F32) {
      if (!rowln) WN_TRY(ln(L.norm_ff, x, t1, M, d, eps, s));
      fS = ffn_fused_try(m, L.ff1, L.ff2, ACT_SILU, s);
      if (fS < 0) return -2;
    }
    if (fS > 0) {
      // partial reduction + residual + norm_final (+ the next layer's norm_ff_macaron)
      if (li + 1 < n_run) {
        const EncLayer& Ln = m->layers[li + 1];
        WN_TRY(ffn_reduce_ln(x, m->ffn_part.as<float>(), fS, L.ff2.b, 0.5f, L.norm_final.w,
                             L.norm_final.b, Ln.norm_ff_mac.w, Ln.norm_ff_mac.b, t1, M, d, eps,
                             1, s));
      } else {
        WN_TRY(ffn_reduce_ln(x, m->ffn_part.as<float>(), fS, L.ff2.b, 0.5f, L.norm_final.w,
                             L.norm_final.b, nullptr, nullptr, nullptr, M, d, eps, 2, s));
      }
      continue;
    }
    WN_TRY(ffn_module(m, L.norm_ff, L.ff1, L.ff2, ACT_SILU, 0.5f,
                      !h16 && t_gemm_prec == PREC_F32, h16, s));
    if (li + 1 < n_run && !(t_mx && h16)) {
      const EncLayer& Ln = m->layers[li + 1];
      WN_TRY(layernorm2(x, L.norm_final.w, L.norm_final.b, Ln.norm_ff_mac.w,
                        Ln.norm_ff_mac.b, x, t1, M, d, eps, s, h16));
    } else {
      WN_TRY(ln(L.norm_final, x, x, M, d, eps, s));
    }
  }
  WN_TRY(m->enc.ensure((size_t)std::max(M, 1) * d * sizeof(float)));
  if (m->dbg_skip_after_norm) {
    WN_HIP(hipMemcpyAsync(m->enc.p, x, (size_t)M * d * sizeof(float),
                          hipMemcpyDeviceToDevice, s));
    return 0;
  }
  WN_TRY(ln(m->after_norm, x, m->enc.as<float>(), M, d, eps, s));
  return 0;
}

// The Conformer layers over ONE chunk of R frames of n_sess streaming sessions with their
// caches (ConformerEncoderLayer.forward with att_cache / cnn_cache, encoder_layer.py:
// 188-265, driven by BaseEncoder.forward_chunk, encoder.py:246-285; batched formulation:
// wenet/bin/export_onnx_gpu.py:83-232).  Same kernels as encoder_layers on n_sess * R rows;
// session b's attention sees its [cache | chunk] keys (ragged: cache lengths differ) with
// the position rows offset_b - t1_b ..., its causal convolution sees its cached left
// context.  All masks are the all-ones fakes of forward_chunk.
int encoder_layers_chunk(wn_model* m, int n_sess, int R, const int* offsets,
                         std::vector<ChunkSess>& sess, float* out, hipStream_t s) {
  const wn_config& c = m->cfg;
  const int d = c.d_model, H = c.n_heads, M = n_sess * R;
  const int lorder = c.causal ? c.cnn_kernel - 1 : 0;
  const int LR = lorder + R;
  float* x = m->x.as<float>();
  float* t1 = m->t1.as<float>();
  float* t2 = m->t2.as<float>();
  float* hb = m->hbuf.as<float>();
  float* qkv = m->qkv.as<float>();
  const float eps = c.norm_eps;
  // descriptors: attention (queries R per session, ragged keys), conv (LR rows per session)
  std::vector<int> qoff(n_sess), qlen(n_sess, R), kvoff(n_sess), kvlen(n_sess), poff(n_sess),
      coff(n_sess), clen(n_sess, LR), rowutt((size_t)n_sess * LR);
  std::vector<int64_t> pw2_rows((size_t)M);
  int total_kv = 0, max_tk = 0;
  for (int b = 0; b < n_sess; ++b) {
    const int tk = sess[b].t1 + R;
    qoff[b] = b * R;
    kvoff[b] = total_kv; kvlen[b] = tk;
    sess[b].kv_off = total_kv;
    total_kv += tk; max_tk = std::max(max_tk, tk);
    // key j of this call sits at position offset - t1 + j   encoder.py:256-257
    poff[b] = offsets[b] - sess[b].t1;
    coff[b] = b * LR;
    for (int r = 0; r < LR; ++r) rowutt[(size_t)b * LR + r] = b;
    for (int r = 0; r < R; ++r) pw2_rows[(size_t)b * R + r] = ((int64_t)b * LR + lorder + r) * d;
  }
  WN_TRY(m->ck_kv.ensure((size_t)total_kv * 2 * d * sizeof(float)));
  WN_TRY(m->ck_xext.ensure((size_t)n_sess * LR * d * sizeof(float)));
  WN_TRY(m->ck_glu.ensure((size_t)n_sess * LR * d * sizeof(float)));
  WN_TRY(m->stage.begin((size_t)(n_sess * (LR + 8) + 64) * sizeof(int) +
                        (size_t)M * sizeof(int64_t) + n_sess * sizeof(ChunkSess) + 4096));
  WN_TRY(upload_desc(m, m->ck_desc, qoff, s));
  WN_TRY(upload_desc(m, m->r_qlen, qlen, s));
  WN_TRY(upload_desc(m, m->r_kvoff, kvoff, s));
  WN_TRY(upload_desc(m, m->r_kvlen, kvlen, s));
  WN_TRY(upload_desc(m, m->r_pos, poff, s));
  WN_TRY(upload_desc(m, m->r_qoff, coff, s));
  WN_TRY(upload_desc(m, m->r_tgt, clen, s));
  WN_TRY(upload_desc(m, m->ck_rowutt, rowutt, s));
  WN_TRY(m->stage.put(m->d_a_row_off, pw2_rows.data(), pw2_rows.size() * sizeof(int64_t), s));
  WN_TRY(m->stage.put(m->ck_sess, sess.data(), sess.size() * sizeof(ChunkSess), s));
  WN_TRY(m->stage.end(s));
  const ChunkSess* dsess = m->ck_sess.as<ChunkSess>();
  float* kv = m->ck_kv.as<float>();
  float* xext = m->ck_xext.as<float>();
  float* glu = m->ck_glu.as<float>();
  for (int li = 0; li < c.n_layers; ++li) {
    const EncLayer& L = m->layers[li];
    WN_TRY(ln(L.norm_ff_mac, x, t1, M, d, eps, s));
    WN_TRY(linear(L.ffm1, t1, d, hb, c.ffn_dim, M, s, ACT_SILU));
    WN_TRY(linear(L.ffm2, hb, c.ffn_dim, x, d, M, s, ACT_NONE, x, d, 0.5f));
    // attention over [cached | new] keys             attention.py:180-245,364-438
    WN_TRY(ln(L.norm_mha, x, t1, M, d, eps, s));
    WN_TRY(linear(L.qkv, t1, d, qkv, 3 * d, M, s));
    WN_TRY(chunk_kv_assemble(dsess, n_sess, li, max_tk, qkv, R, H, kv, s));
    AttnArgs a;
    a.Q = qkv; a.ldq = 3 * d;
    a.K = kv; a.V = kv + d; a.ldk = a.ldv = 2 * d;
    a.P = L.pos_tab; a.ldp = d; a.p_off = m->r_pos.as<int>();
    a.bias_u = L.bias_u; a.bias_v = L.bias_v;
    a.O = t2; a.ldo = d;
    a.q_off = m->ck_desc.as<int>(); a.q_len = m->r_qlen.as<int>();
    a.kv_off = m->r_kvoff.as<int>(); a.kv_len = m->r_kvlen.as<int>();
    a.n_seq = n_sess; a.n_heads = H; a.max_q_len = R;
    a.mask_mode = 0;
    a.scale = 1.0f / sqrtf(64.0f);
    WN_TRY(attention(a, s));
    WN_TRY(linear(L.out, t2, d, x, d, M, s, ACT_NONE, x, d));
    // convolution module with its left-context cache  convolution.py:98-153
    WN_TRY(ln(L.norm_conv, x, t1, M, d, eps, s));
    if (lorder > 0) {
      WN_TRY(chunk_conv_input(dsess, n_sess, li, t1, R, d, lorder, xext, s));
      WN_TRY(linear(L.pw1, xext, d, glu, d, n_sess * LR, s, ACT_NONE, nullptr, 0, 1.0f, true));
    } else {
      WN_TRY(linear(L.pw1, t1, d, glu, d, M, s, ACT_NONE, nullptr, 0, 1.0f, true));
    }
    DwConvArgs dw;
    dw.x = glu; dw.ldx = d; dw.wt = L.dw_wt; dw.bias = L.dw_b; dw.cpad = L.cpad;
    dw.ln_w = L.conv_norm.w; dw.ln_b = L.conv_norm.b; dw.norm_mode = c.cnn_norm;
    dw.y = xext; dw.ldy = d;
    dw.row_utt = m->ck_rowutt.as<int>(); dw.off = m->r_qoff.as<int>();
    dw.len = m->r_tgt.as<int>();
    dw.M = n_sess * LR; dw.D = d; dw.K = c.cnn_kernel; dw.causal = c.causal;
    dw.t_max = LR; dw.eps = 1e-5f;
    WN_TRY(dwconv_ln_silu(dw, s));
    {
      // pointwise_conv2 on the chunk rows of every session (rows lorder.. of its segment)
      GemmArgs g;
      g.A = xext; g.W = L.pw2.w; g.bias = L.pw2.b; g.C = x; g.resid = x;
      g.M = M; g.N = d; g.K = d; g.lda = d; g.ldc = d; g.ldr = d;
      g.a_row_off = m->d_a_row_off.as<int64_t>(); g.conv_C = d;
      WN_TRY(gemm_f32(g, s));
    }
    WN_TRY(ln(L.norm_ff, x, t1, M, d, eps, s));
    WN_TRY(linear(L.ff1, t1, d, hb, c.ffn_dim, M, s, ACT_SILU));
    WN_TRY(linear(L.ff2, hb, c.ffn_dim, x, d, M, s, ACT_NONE, x, d, 0.5f));
    WN_TRY(ln(L.norm_final, x, x, M, d, eps, s));
  }
  WN_TRY(ln(m->after_norm, x, out, M, d, eps, s));
  return 0;
}

// TransformerEncoder (Whisper style): x += MHA(LN(x)); x += FFN(LN(x)); final LN
// (encoder_layer.py:94-127, encoder.py:176-181).
int transformer_layers(wn_model* m, hipStream_t s) {
  const wn_config& c = m->cfg;
  const int d = c.d_model, M = m->rows;
  float* x = m->x.as<float>();
  float* t1 = m->t1.as<float>();
  float* t2 = m->t2.as<float>();
  float* hb = m->hbuf.as<float>();
  float* qkv = m->qkv.as<float>();
  const float eps = c.norm_eps;
  int max_len = 0;
  for (int b = 0; b < m->B; ++b) max_len = std::max(max_len, m->len[b]);
  const int act = c.activation == 1 ? ACT_GELU : ACT_SILU;
  const bool h16 = bf16_store_active();  // t1, t2, hb hold bf16 (see there)
  const int n_run = m->dbg_layers >= 0 ? std::min(m->dbg_layers, c.n_layers)
                                      : c.n_layers;
  for (int li = 0; li < n_run; ++li) {
    const TfLayer& L = m->tf_layers[li];
    WN_TRY(ln(L.n1, x, t1, M, d, eps, s, h16));
    // bf16-storage form: Q | K | V leave the GEMM as bf16 (the attention kernel rounds
    // them to bf16 first thing anyway): half the GEMM's store and the attention's stream
    const bool q16 = h16 && g_qkv_bf16 != 0;
    WN_TRY(linear(L.qkv, t1, d, qkv, 3 * d, M, s, ACT_NONE, nullptr, 0, 1.0f, false,
                  h16, q16));
    AttnArgs a;
    if (q16) {
      const __bf16* qh = reinterpret_cast<const __bf16*>(qkv);
      a.Q = reinterpret_cast<const float*>(qh);
      a.K = reinterpret_cast<const float*>(qh + d);
      a.V = reinterpret_cast<const float*>(qh + 2 * d);
      a.qkv_bf16 = true;
    } else {
      a.Q = qkv; a.K = qkv + d; a.V = qkv + 2 * d;
    }
    a.ldq = a.ldk = a.ldv = 3 * d;
    a.O = t2; a.ldo = d; a.o_bf16 = h16;
    a.q_off = a.kv_off = m->d_off.as<int>();
    a.q_len = a.kv_len = m->d_len.as<int>();
    a.n_seq = m->B; a.n_heads = c.n_heads; a.max_q_len = max_len;
    a.mask_mode = 0;
    a.scale = 1.0f / sqrtf(64.0f);
    WN_TRY(attention(a, s));
    WN_TRY(linear(L.out, t2, d, x, d, M, s, ACT_NONE, x, d, 1.0f, false, h16));
    WN_TRY(ffn_module(m, L.n2, L.ff1, L.ff2, act, 1.0f, false, h16, s));
  }
  WN_TRY(m->enc.ensure((size_t)std::max(M, 1) * d * sizeof(float)));
  if (m->dbg_skip_after_norm) {
    WN_HIP(hipMemcpyAsync(m->enc.p, x, (size_t)M * d * sizeof(float),
                          hipMemcpyDeviceToDevice, s));
    return 0;
  }
  WN_TRY(ln(m->after_norm, x, m->enc.as<float>(), M, d, eps, s));
  return 0;
}

// Conv1dSubsampling2 + WhisperPositionalEncoding + the layers above
// (subsampling.py:117-171, embedding.py:150-164, encoder.py:122-181).
int encode_transformer(wn_model* m, const float* feats_dev,
                       const int32_t* feat_lens_host, int B, int T,
                       float* enc_out_dev, int32_t* enc_lens_host, hipStream_t s) {
  const wn_config& c = m->cfg;
  const int d = c.d_model, F = c.feat_dim;
  // output frames: conv(k3, s2, p1) keeps floor((T-1)/2)+1; the mask keeps
  // x_mask[:, :, (T+1)%2::2] (subsampling.py:171): frame t' is valid iff
  // 2t' + (T+1)%2 < len
  const int Tp = (T - 1) / 2 + 1;
  const int par = (T + 1) % 2;
  WN_CHECK(Tp <= c.max_pos, "utterance longer than the positional table");
  std::vector<int> seg(B), off2(B), len2(B), lens(B);
  int rows_pad = 0, M = 0;
  std::vector<int> zero_rows;
  for (int b = 0; b < B; ++b) {
    const int L = feat_lens_host[b];
    WN_CHECK(L >= 0 && L <= T, "wn_encode: feature length out of range");
    lens[b] = L;
    seg[b] = rows_pad;
    rows_pad += L + 3;
    const int l2 = L > par ? (L - par + 1) / 2 : 0;   // #{t' : 2t' + par < L}
    off2[b] = M; len2[b] = l2; M += l2;
    if (enc_lens_host) enc_lens_host[b] = l2;
    zero_rows.push_back(seg[b]);                       // conv2's left zero pad
    // conv1 position L exists in the reference only as a padded frame of a
    // longer batch; when the utterance fills the tensor it is conv2's right
    // zero pad instead
    if (L == T) zero_rows.push_back(seg[b] + 1 + L);
  }
  m->B = B; m->Tp = Tp; m->off = off2; m->len = len2; m->rows = M;
  m->ctc_valid = false;
  m->mem_cache_valid = false;
  if (M == 0) {
    WN_TRY(m->stage.begin((size_t)B * 16 + 1024));
    WN_TRY(upload_desc(m, m->d_off, off2, s));
    WN_TRY(upload_desc(m, m->d_len, len2, s));
    WN_TRY(m->stage.end(s));
  } else {
    std::vector<int64_t> a_off((size_t)M);
    std::vector<int> row_t((size_t)M);
    for (int b = 0; b < B; ++b)
      for (int t = 0; t < len2[b]; ++t) {
        // conv2 output t' reads conv1 positions 2t'-1 .. 2t'+1 = c1pad rows
        // seg + 2t' .. seg + 2t' + 2
        a_off[off2[b] + t] = (int64_t)(seg[b] + 2 * t) * d;
        row_t[off2[b] + t] = t;
      }
    WN_TRY(m->stage.begin((size_t)M * 16 + (size_t)B * 64 * 5 + zero_rows.size() * 4 +
                          64 * 12 + 4096));
    WN_TRY(upload_desc(m, m->d_off, off2, s));
    WN_TRY(upload_desc(m, m->d_len, len2, s));
    {
      std::vector<int> row_utt(M, -1);
      for (int b = 0; b < B; ++b)
        for (int t = 0; t < len2[b]; ++t) row_utt[off2[b] + t] = b;
      WN_TRY(upload_desc(m, m->d_row_utt, row_utt, s));
    }
    WN_TRY(upload_desc(m, m->d_off1, seg, s));
    WN_TRY(upload_desc(m, m->d_len1, lens, s));
    WN_TRY(upload_desc(m, m->d_row_t, row_t, s));
    WN_TRY(upload_desc(m, m->d_zero_rows, zero_rows, s));
    WN_TRY(m->stage.put(m->d_a_row_off, a_off.data(), a_off.size() * sizeof(int64_t), s));
    WN_TRY(m->stage.end(s));
    const int K1 = m->tconv1.in;  // 3F rounded up to 32 (zero weights)
    WN_TRY(m->xpad.ensure(((size_t)rows_pad * F + K1 + 64) * sizeof(float)));
    WN_TRY(m->c1.ensure(((size_t)rows_pad + 2) * d * sizeof(float)));
    WN_TRY(m->x.ensure((size_t)M * d * sizeof(float)));
    WN_TRY(m->t1.ensure((size_t)M * d * sizeof(float)));
    WN_TRY(m->t2.ensure((size_t)M * d * sizeof(float)));
    WN_TRY(m->pos_rows.ensure((size_t)M * d * sizeof(float)));
    WN_TRY(m->hbuf.ensure((size_t)M * c.ffn_dim * sizeof(float)));
    WN_TRY(m->qkv.ensure((size_t)M * 3 * d * sizeof(float)));
    int max_len = 0;
    for (int b = 0; b < B; ++b) max_len = std::max(max_len, lens[b]);
    // the K padding of the last rows reads a few floats past the data: keep
    // them finite (they meet zero weights)
    WN_HIP(hipMemsetAsync(m->xpad.as<float>() + (size_t)rows_pad * F, 0,
                          (size_t)(K1 + 64) * sizeof(float), s));
    hipLaunchKernelGGL(pad_feats_kernel, dim3(max_len + 3, B), dim3(64), 0, s,
                       feats_dev, T, F, m->d_off1.as<int>(), m->d_len1.as<int>(),
                       m->cmvn_mean, m->cmvn_istd, m->xpad.as<float>());
    WN_HIP(hipGetLastError());
    // conv1 (k3, pad 1) + GELU: output row r = taps at xpad rows r, r+1, r+2
    // -> c1pad row r + 1 (row seg_b is the zero pad in front of utterance b)
    GemmArgs g1;
    g1.A = m->xpad.as<float>(); g1.W = m->tconv1.w; g1.bias = m->tconv1.b;
    g1.C = m->c1.as<float>() + d; g1.M = rows_pad - 2; g1.N = d; g1.K = K1;
    g1.lda = F; g1.ldc = d; g1.act = ACT_GELU;
    WN_CHECK(F % 4 == 0, "conv1d2 front end: feature dim must be a multiple of 4");
    WN_TRY(gemm_f32(g1, s));
    hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)zero_rows.size()), dim3(64), 0,
                       s, m->c1.as<float>(), d / 4, m->d_zero_rows.as<int>(),
                       (int)zero_rows.size());
    WN_HIP(hipGetLastError());
    // positional rows pe[t'] (xscale = 1, embedding.py:156)
    WN_TRY(copy_rows(m->pe, d, m->d_row_t.as<int>(), m->pos_rows.as<float>(), d,
                     nullptr, M, d, s));
    // conv2 (k3, stride 2, pad 1) + GELU, + pe: gathered rows of c1pad
    GemmArgs g2;
    g2.A = m->c1.as<float>(); g2.W = m->tconv2.w; g2.bias = m->tconv2.b;
    g2.C = m->x.as<float>(); g2.M = M; g2.N = d; g2.K = 3 * d; g2.ldc = d;
    g2.act = ACT_GELU; g2.resid = m->pos_rows.as<float>(); g2.ldr = d;
    g2.a_row_off = m->d_a_row_off.as<int64_t>();
    g2.conv_C = 3 * d; g2.conv_sy = 0; g2.conv_sx = 0;
    WN_TRY(gemm_f32(g2, s));
    WN_TRY(transformer_layers(m, s));
  }
  if (enc_out_dev) {
    if (M > 0) {
      hipLaunchKernelGGL(scatter_padded_kernel, dim3(Tp, B), dim3(64), 0, s,
                         m->enc.as<float>(), d, m->d_off.as<int>(),
                         m->d_len.as<int>(), Tp, d / 4, enc_out_dev);
      WN_HIP(hipGetLastError());
    } else if (Tp > 0) {
      WN_HIP(hipMemsetAsync(enc_out_dev, 0, (size_t)B * Tp * d * sizeof(float), s));
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------
// weight ingestion
struct HostStage {
  std::vector<float> data;
  std::map<std::string, std::pair<size_t, size_t>> at;  // name -> (offset, n)
  void add(const std::string& name, const float* p, size_t n) {
    size_t o = (data.size() + 63) / 64 * 64;
    data.resize(o + n);
    memcpy(data.data() + o, p, n * sizeof(float));
    at[name] = {o, n};
  }
  float* alloc(const std::string& name, size_t n) {
    size_t o = (data.size() + 63) / 64 * 64;
    data.resize(o + n, 0.f);
    at[name] = {o, n};
    return data.data() + o;  // valid until the next add/alloc
  }
};

struct Src {
  std::map<std::string, std::pair<const float*, int64_t>> t;
  const float* get(const std::string& n, int64_t numel) const {
    auto it = t.find(n);
    if (it == t.end()) { set_error("missing weight: " + n); return nullptr; }
    if (numel >= 0 && it->second.second != numel) {
      set_error("weight " + n + " has " + std::to_string(it->second.second) +
                " elements, expected " + std::to_string(numel));
      return nullptr;
    }
    return it->second.first;
  }
  bool has(const std::string& n) const { return t.count(n) != 0; }
};

#define WN_GET(var, name, numel)                 \
  const float* var = src.get((name), (numel));   \
  if (!var) return -3;

int stage_linear(const Src& src, HostStage& hs, const std::string& pfx, int out,
                 int in, bool bias = true) {
  WN_GET(w, pfx + ".weight", (int64_t)out * in);
  hs.add(pfx + ".weight", w, (size_t)out * in);
  if (bias) {
    WN_GET(b, pfx + ".bias", out);
    hs.add(pfx + ".bias", b, out);
  }
  return 0;
}
int stage_norm(const Src& src, HostStage& hs, const std::string& pfx, int n) {
  WN_GET(w, pfx + ".weight", n);
  WN_GET(b, pfx + ".bias", n);
  hs.add(pfx + ".weight", w, n);
  hs.add(pfx + ".bias", b, n);
  return 0;
}
// fuse several Linear layers along the output dimension
int stage_fused(const Src& src, HostStage& hs, const std::string& name,
                const std::vector<std::string>& parts, int out_each, int in) {
  std::vector<float> w((size_t)parts.size() * out_each * in),
      b((size_t)parts.size() * out_each);
  for (size_t i = 0; i < parts.size(); ++i) {
    WN_GET(pw, parts[i] + ".weight", (int64_t)out_each * in);
    WN_GET(pb, parts[i] + ".bias", out_each);
    memcpy(w.data() + i * out_each * in, pw, sizeof(float) * out_each * in);
    memcpy(b.data() + i * out_each, pb, sizeof(float) * out_each);
  }
  hs.add(name + ".weight", w.data(), w.size());
  hs.add(name + ".bias", b.data(), b.size());
  return 0;
}

int stage_decoder(const Src& src, HostStage& hs, const std::string& pfx,
                  int nlayers, const wn_config& c) {
  const int d = c.d_model, V = c.vocab;
  WN_GET(emb, pfx + ".embed.0.weight", (int64_t)V * d);
  hs.add(pfx + ".embed", emb, (size_t)V * d);
  WN_TRY(stage_norm(src, hs, pfx + ".after_norm", d));
  WN_TRY(stage_linear(src, hs, pfx + ".output_layer", V, d));
  for (int j = 0; j < nlayers; ++j) {
    const std::string p = pfx + ".decoders." + std::to_string(j);
    WN_TRY(stage_fused(src, hs, p + ".self_qkv",
                       {p + ".self_attn.linear_q", p + ".self_attn.linear_k",
                        p + ".self_attn.linear_v"}, d, d));
    WN_TRY(stage_linear(src, hs, p + ".self_attn.linear_out", d, d));
    WN_TRY(stage_linear(src, hs, p + ".src_attn.linear_q", d, d));
    WN_TRY(stage_fused(src, hs, p + ".src_kv",
                       {p + ".src_attn.linear_k", p + ".src_attn.linear_v"}, d,
                       d));
    WN_TRY(stage_linear(src, hs, p + ".src_attn.linear_out", d, d));
    WN_TRY(stage_linear(src, hs, p + ".feed_forward.w_1", c.dec_ffn_dim, d));
    WN_TRY(stage_linear(src, hs, p + ".feed_forward.w_2", d, c.dec_ffn_dim));
    for (const char* n : {"norm1", "norm2", "norm3"})
      WN_TRY(stage_norm(src, hs, p + "." + n, d));
  }
  return 0;
}

}  // namespace

// ===========================================================================
extern "C" {

const char* wn_last_error(void) { return g_error.c_str(); }
const char* wn_version(void) { return "wenet_amd 0.1 (gfx950, fp32 MFMA)"; }

int wn_model_create(const wn_config* cfg, const wn_tensor* weights,
                    int32_t n_weights, int32_t device, wn_model** out) {
  WN_CHECK(cfg && weights && out, "wn_model_create: null argument");
  const wn_config& c = *cfg;
  WN_CHECK(c.d_model % 64 == 0 && c.n_heads > 0 && c.d_model / c.n_heads == 64,
           "d_model / n_heads must be 64 (all reference Conformer configs)");
  WN_CHECK(c.dec_layers == 0 || c.dec_heads == 0 || c.d_model / c.dec_heads == 64,
           "decoder head dim must be 64");
  WN_CHECK(c.ffn_dim % 32 == 0 && c.feat_dim >= 7 && c.feat_dim <= 128,
           "unsupported ffn_dim / feat_dim");
  const bool tf = c.encoder_type == 1;
  WN_CHECK(c.encoder_type == 0 || c.encoder_type == 1, "unknown encoder_type");
  WN_CHECK(tf ? c.input_layer == 1 : c.input_layer == 0,
           "supported pairs: conformer + conv2d, transformer + conv1d2");
  WN_CHECK(tf || (c.cnn_kernel >= 1 && (c.causal || c.cnn_kernel % 2 == 1)),
           "cnn_module_kernel must be odd for a non-causal conv module");
  WN_HIP(hipSetDevice(device));
  std::unique_ptr<wn_model> m(new wn_model());
  m->cfg = c;
  m->device = device;
  Src src;
  for (int i = 0; i < n_weights; ++i)
    src.t[weights[i].name] = {weights[i].data, weights[i].numel};

  const int d = c.d_model, F = c.ffn_dim, V = c.vocab, K = c.cnn_kernel;
  const int F2 = m->F2();
  HostStage hs;
  if (c.has_cmvn) {
    WN_GET(mean, "encoder.global_cmvn.mean", c.feat_dim);
    WN_GET(istd, "encoder.global_cmvn.istd", c.feat_dim);
    hs.add("cmvn.mean", mean, c.feat_dim);
    hs.add("cmvn.istd", istd, c.feat_dim);
  }
  const int Fin = c.feat_dim;
  const int K1 = cdiv(3 * Fin, 32) * 32;  // conv1d K, padded with zero weights
  if (tf) {
    // Conv1d (n, c, tap) -> [n][tap * C + c] (the taps of one output frame are
    // three consecutive channels-last input rows)
    WN_GET(w0, "encoder.embed.conv.0.weight", (int64_t)d * Fin * 3);
    WN_GET(b0, "encoder.embed.conv.0.bias", d);
    float* t = hs.alloc("tconv1.w", (size_t)d * K1);
    for (int n = 0; n < d; ++n)
      for (int ch = 0; ch < Fin; ++ch)
        for (int k = 0; k < 3; ++k)
          t[(size_t)n * K1 + (size_t)k * Fin + ch] = w0[((size_t)n * Fin + ch) * 3 + k];
    hs.add("tconv1.b", b0, d);
    WN_GET(w2, "encoder.embed.conv.2.weight", (int64_t)d * d * 3);
    WN_GET(b2, "encoder.embed.conv.2.bias", d);
    t = hs.alloc("tconv2.w", (size_t)d * 3 * d);
    for (int n = 0; n < d; ++n)
      for (int ch = 0; ch < d; ++ch)
        for (int k = 0; k < 3; ++k)
          t[(size_t)n * 3 * d + (size_t)k * d + ch] = w2[((size_t)n * d + ch) * 3 + k];
    hs.add("tconv2.b", b2, d);
  } else {  // conv1 (d,1,3,3) -> [tap][c]
    WN_GET(w0, "encoder.embed.conv.0.weight", (int64_t)d * 9);
    WN_GET(b0, "encoder.embed.conv.0.bias", d);
    float* t = hs.alloc("conv1.w", (size_t)9 * d);
    for (int ch = 0; ch < d; ++ch)
      for (int k = 0; k < 9; ++k) t[k * d + ch] = w0[ch * 9 + k];
    hs.add("conv1.b", b0, d);
    // conv2 (n, c, ky, kx) -> [n][(ky*3+kx)*d + c]
    WN_GET(w2, "encoder.embed.conv.2.weight", (int64_t)d * d * 9);
    WN_GET(b2, "encoder.embed.conv.2.bias", d);
    t = hs.alloc("conv2.w", (size_t)d * 9 * d);
    for (int n = 0; n < d; ++n)
      for (int ch = 0; ch < d; ++ch)
        for (int k = 0; k < 9; ++k)
          t[(size_t)n * 9 * d + (size_t)k * d + ch] =
              w2[((size_t)n * d + ch) * 9 + k];
    hs.add("conv2.b", b2, d);
    // out Linear(d*F2 -> d): input index c*F2+f  ->  f*d+c
    WN_GET(wo, "encoder.embed.out.0.weight", (int64_t)d * d * F2);
    WN_GET(bo, "encoder.embed.out.0.bias", d);
    t = hs.alloc("sub_out.w", (size_t)d * d * F2);
    for (int n = 0; n < d; ++n)
      for (int ch = 0; ch < d; ++ch)
        for (int f = 0; f < F2; ++f)
          t[(size_t)n * d * F2 + (size_t)f * d + ch] =
              wo[(size_t)n * d * F2 + (size_t)ch * F2 + f];
    hs.add("sub_out.b", bo, d);
  }
  {  // positional table: the `pe` buffer (embedding.py:47-56)
    float* t = hs.alloc("pe", (size_t)c.max_pos * d);
    WN_CHECK(!tf || src.has("encoder.embed.pos_enc.pe"),
             "transformer encoder: encoder.embed.pos_enc.pe is required");
    if (src.has("encoder.embed.pos_enc.pe")) {
      WN_GET(pe, "encoder.embed.pos_enc.pe", (int64_t)c.max_pos * d);
      memcpy(t, pe, sizeof(float) * c.max_pos * d);
    } else {
      for (int pos = 0; pos < c.max_pos; ++pos)
        for (int i = 0; i < d; i += 2) {
          const float div = expf((float)i * -(logf(10000.0f) / (float)d));
          t[(size_t)pos * d + i] = sinf((float)pos * div);
          t[(size_t)pos * d + i + 1] = cosf((float)pos * div);
        }
    }
  }
  // ---- fbank tables (runtime/core/frontend/fbank.h:91-163) -----------------
  std::vector<int> mel_start(c.feat_dim), mel_len(c.feat_dim), mel_off(c.feat_dim);
  {
    float* win = hs.alloc("fbank.window", 400);
    const double a = 2.0 * M_PI / 399.0;
    for (int i = 0; i < 400; ++i) win[i] = (float)pow(0.5 - 0.5 * cos(a * i), 0.85);
    float* tw = hs.alloc("fbank.twiddle", 512);
    for (int k = 0; k < 256; ++k) {
      tw[2 * k] = (float)cos(2.0 * M_PI * k / 512.0);
      tw[2 * k + 1] = (float)-sin(2.0 * M_PI * k / 512.0);
    }
    auto mel = [](float f) { return 1127.0f * logf(1.0f + f / 700.0f); };
    const int nbins = c.feat_dim, nfft_bins = 256;
    const float bin_w = 16000.0f / 512.0f;
    const float mlo = mel(20.0f), mhi = mel(8000.0f);
    const float delta = (mhi - mlo) / (float)(nbins + 1);
    std::vector<float> wts;
    for (int b = 0; b < nbins; ++b) {
      const float left = mlo + b * delta, center = mlo + (b + 1) * delta,
                  right = mlo + (b + 2) * delta;
      int first = -1, last = -1;
      std::vector<float> row(nfft_bins, 0.f);
      for (int i = 0; i < nfft_bins; ++i) {
        const float mf = mel(bin_w * i);
        if (mf > left && mf < right) {
          row[i] = mf <= center ? (mf - left) / (center - left)
                                : (right - mf) / (right - center);
          if (first < 0) first = i;
          last = i;
        }
      }
      if (first < 0) { m->fbank_ok = false; first = last = 0; }  // e.g. 128 bins
      mel_start[b] = first; mel_len[b] = last + 1 - first; mel_off[b] = (int)wts.size();
      for (int i = first; i <= last; ++i) wts.push_back(row[i]);
    }
    hs.add("fbank.mel_w", wts.data(), wts.size());
  }
  WN_TRY(stage_norm(src, hs, "encoder.after_norm", d));
  const bool has_ctc = src.has("ctc.ctc_lo.weight");
  if (has_ctc) WN_TRY(stage_linear(src, hs, "ctc.ctc_lo", V, d));
  for (int i = 0; tf && i < c.n_layers; ++i) {
    const std::string p = "encoder.encoders." + std::to_string(i);
    WN_TRY(stage_norm(src, hs, p + ".norm1", d));
    WN_TRY(stage_norm(src, hs, p + ".norm2", d));
    {  // fused QKV; Whisper's linear_k has no bias (attention.py:29-75)
      std::vector<float> w((size_t)3 * d * d), b((size_t)3 * d, 0.f);
      const char* parts[3] = {"linear_q", "linear_k", "linear_v"};
      for (int j = 0; j < 3; ++j) {
        const std::string q = p + ".self_attn." + parts[j];
        WN_GET(pw, q + ".weight", (int64_t)d * d);
        memcpy(w.data() + (size_t)j * d * d, pw, sizeof(float) * d * d);
        if (j != 1 || c.key_bias) {
          WN_GET(pb, q + ".bias", d);
          memcpy(b.data() + (size_t)j * d, pb, sizeof(float) * d);
        }
      }
      hs.add(p + ".qkv.weight", w.data(), w.size());
      hs.add(p + ".qkv.bias", b.data(), b.size());
    }
    WN_TRY(stage_linear(src, hs, p + ".self_attn.linear_out", d, d));
    WN_TRY(stage_linear(src, hs, p + ".feed_forward.w_1", F, d));
    WN_TRY(stage_linear(src, hs, p + ".feed_forward.w_2", d, F));
  }
  for (int i = 0; !tf && i < c.n_layers; ++i) {
    const std::string p = "encoder.encoders." + std::to_string(i);
    for (const char* n : {"norm_ff_macaron", "norm_mha", "norm_conv", "norm_ff",
                          "norm_final"})
      WN_TRY(stage_norm(src, hs, p + "." + n, d));
    if (c.cnn_norm == 0) {
      WN_TRY(stage_norm(src, hs, p + ".conv_module.norm", d));
    } else {
      // eval-mode BatchNorm1d (convolution.py:77-81,139-143) as a per-channel
      // affine: y = x * scale + shift, scale = w / sqrt(running_var + eps),
      // shift = b - running_mean * scale; staged in the norm's weight / bias slots
      const std::string q = p + ".conv_module.norm";
      WN_GET(bw, q + ".weight", d);
      WN_GET(bb, q + ".bias", d);
      WN_GET(bm, q + ".running_mean", d);
      WN_GET(bv, q + ".running_var", d);
      std::vector<float> sc(d), sh(d);
      for (int ch = 0; ch < d; ++ch) {
        const float inv = 1.0f / sqrtf(bv[ch] + c.norm_eps);
        sc[ch] = bw[ch] * inv;
        sh[ch] = bb[ch] - bm[ch] * sc[ch];
      }
      hs.add(q + ".weight", sc.data(), sc.size());
      hs.add(q + ".bias", sh.data(), sh.size());
    }
    for (const char* ff : {"feed_forward_macaron", "feed_forward"}) {
      WN_TRY(stage_linear(src, hs, p + "." + ff + ".w_1", F, d));
      WN_TRY(stage_linear(src, hs, p + "." + ff + ".w_2", d, F));
    }
    WN_TRY(stage_fused(src, hs, p + ".qkv",
                       {p + ".self_attn.linear_q", p + ".self_attn.linear_k",
                        p + ".self_attn.linear_v"}, d, d));
    WN_TRY(stage_linear(src, hs, p + ".self_attn.linear_out", d, d));
    WN_TRY(stage_linear(src, hs, p + ".self_attn.linear_pos", d, d, false));
    WN_GET(bu, p + ".self_attn.pos_bias_u", d);
    WN_GET(bv, p + ".self_attn.pos_bias_v", d);
    hs.add(p + ".pos_bias_u", bu, d);
    hs.add(p + ".pos_bias_v", bv, d);
    {  // pointwise_conv1 (2d, d, 1): rows permuted per 64 as [32 a | 32 gate]
      WN_GET(w1, p + ".conv_module.pointwise_conv1.weight", (int64_t)2 * d * d);
      WN_GET(b1, p + ".conv_module.pointwise_conv1.bias", 2 * d);
      std::vector<float> w((size_t)2 * d * d), b(2 * d), cp(d);
      for (int g = 0; g < d / 32; ++g)
        for (int j = 0; j < 32; ++j) {
          const int ch = g * 32 + j;
          memcpy(&w[(size_t)(g * 64 + j) * d], &w1[(size_t)ch * d],
                 sizeof(float) * d);
          memcpy(&w[(size_t)(g * 64 + 32 + j) * d], &w1[(size_t)(d + ch) * d],
                 sizeof(float) * d);
          b[g * 64 + j] = b1[ch];
          b[g * 64 + 32 + j] = b1[d + ch];
          // GLU of a zero input frame: bias_a * sigmoid(bias_gate)
          cp[ch] = b1[ch] * (1.0f / (1.0f + expf(-b1[d + ch])));
        }
      hs.add(p + ".pw1.weight", w.data(), w.size());
      hs.add(p + ".pw1.bias", b.data(), b.size());
      hs.add(p + ".cpad", cp.data(), cp.size());
    }
    {  // depthwise (d,1,K) -> [K][d]
      WN_GET(wd, p + ".conv_module.depthwise_conv.weight", (int64_t)d * K);
      WN_GET(bd, p + ".conv_module.depthwise_conv.bias", d);
      std::vector<float> w((size_t)K * d);
      for (int ch = 0; ch < d; ++ch)
        for (int k = 0; k < K; ++k) w[(size_t)k * d + ch] = wd[(size_t)ch * K + k];
      hs.add(p + ".dw.weight", w.data(), w.size());
      hs.add(p + ".dw.bias", bd, d);
    }
    WN_TRY(stage_linear(src, hs, p + ".conv_module.pointwise_conv2", d, d));
  }
  const bool has_dec = c.dec_layers > 0;
  if (has_dec) {
    if (c.bidirectional) {
      WN_TRY(stage_decoder(src, hs, "decoder.left_decoder", c.dec_layers, c));
      if (c.dec_r_layers > 0)
        WN_TRY(stage_decoder(src, hs, "decoder.right_decoder", c.dec_r_layers, c));
    } else {
      WN_TRY(stage_decoder(src, hs, "decoder", c.dec_layers, c));
    }
  }
  // ---- upload ---------------------------------------------------------------
  WN_TRY(m->weights->ensure(hs.data.size() * sizeof(float)));
  m->n_weight_elems = (int64_t)hs.data.size();
  WN_HIP(hipMemcpy(m->weights->p, hs.data.data(), hs.data.size() * sizeof(float),
                   hipMemcpyHostToDevice));
  const float* base = m->weights->as<float>();
  for (auto& kv : hs.at) m->w[kv.first] = base + kv.second.first;
  auto W = [&](const std::string& n) { return m->w.at(n); };
  auto LIN = [&](const std::string& p, int o, int i, bool bias = true) {
    Linear l; l.w = W(p + ".weight"); l.b = bias ? W(p + ".bias") : nullptr;
    l.out = o; l.in = i; return l;
  };
  auto NORM = [&](const std::string& p) {
    Norm n; n.w = W(p + ".weight"); n.b = W(p + ".bias"); return n;
  };
  if (c.has_cmvn) { m->cmvn_mean = W("cmvn.mean"); m->cmvn_istd = W("cmvn.istd"); }
  if (tf) {
    m->tconv1.w = W("tconv1.w"); m->tconv1.b = W("tconv1.b");
    m->tconv1.out = d; m->tconv1.in = K1;
    m->tconv2.w = W("tconv2.w"); m->tconv2.b = W("tconv2.b");
    m->tconv2.out = d; m->tconv2.in = 3 * d;
  } else {
    m->conv1_w = W("conv1.w"); m->conv1_b = W("conv1.b");
    m->conv2.w = W("conv2.w"); m->conv2.b = W("conv2.b");
    m->conv2.out = d; m->conv2.in = 9 * d;
    m->sub_out.w = W("sub_out.w"); m->sub_out.b = W("sub_out.b");
    m->sub_out.out = d; m->sub_out.in = d * F2;
  }
  m->pe = W("pe");
  m->fb_window = W("fbank.window"); m->fb_twiddle = W("fbank.twiddle");
  m->fb_mel_w = W("fbank.mel_w");
  {
    std::vector<int> tab;
    tab.insert(tab.end(), mel_start.begin(), mel_start.end());
    tab.insert(tab.end(), mel_len.begin(), mel_len.end());
    tab.insert(tab.end(), mel_off.begin(), mel_off.end());
    WN_TRY(m->fb_tab_i->ensure(tab.size() * sizeof(int)));
    WN_HIP(hipMemcpy(m->fb_tab_i->p, tab.data(), tab.size() * sizeof(int),
                     hipMemcpyHostToDevice));
  }
  m->after_norm = NORM("encoder.after_norm");
  if (has_ctc) m->ctc = LIN("ctc.ctc_lo", V, d);
  if (tf) {
    m->tf_layers.resize(c.n_layers);
    for (int i = 0; i < c.n_layers; ++i) {
      const std::string p = "encoder.encoders." + std::to_string(i);
      TfLayer& L = m->tf_layers[i];
      L.n1 = NORM(p + ".norm1"); L.n2 = NORM(p + ".norm2");
      L.qkv = LIN(p + ".qkv", 3 * d, d);
      L.out = LIN(p + ".self_attn.linear_out", d, d);
      L.ff1 = LIN(p + ".feed_forward.w_1", F, d);
      L.ff2 = LIN(p + ".feed_forward.w_2", d, F);
    }
  } else {
    m->layers.resize(c.n_layers);
    WN_TRY(m->pos_tabs->ensure((size_t)c.n_layers * c.max_pos * d * sizeof(float)));
  }
  for (int i = 0; !tf && i < c.n_layers; ++i) {
    const std::string p = "encoder.encoders." + std::to_string(i);
    EncLayer& L = m->layers[i];
    L.norm_ff_mac = NORM(p + ".norm_ff_macaron");
    L.norm_mha = NORM(p + ".norm_mha");
    L.norm_conv = NORM(p + ".norm_conv");
    L.norm_ff = NORM(p + ".norm_ff");
    L.norm_final = NORM(p + ".norm_final");
    L.conv_norm = NORM(p + ".conv_module.norm");
    L.ffm1 = LIN(p + ".feed_forward_macaron.w_1", F, d);
    L.ffm2 = LIN(p + ".feed_forward_macaron.w_2", d, F);
    L.ff1 = LIN(p + ".feed_forward.w_1", F, d);
    L.ff2 = LIN(p + ".feed_forward.w_2", d, F);
    L.qkv = LIN(p + ".qkv", 3 * d, d);
    L.out = LIN(p + ".self_attn.linear_out", d, d);
    L.pw1 = LIN(p + ".pw1", 2 * d, d);
    L.pw2 = LIN(p + ".conv_module.pointwise_conv2", d, d);
    L.bias_u = W(p + ".pos_bias_u");
    L.bias_v = W(p + ".pos_bias_v");
    L.pos_w = W(p + ".self_attn.linear_pos.weight");
    L.dw_wt = W(p + ".dw.weight");
    L.dw_b = W(p + ".dw.bias");
    L.cpad = W(p + ".cpad");
    // p = linear_pos(pos_emb) depends on weights only (attention.py:395-396):
    // project the whole table once instead of per batch and layer.
    L.pos_tab = m->pos_tabs->as<float>() + (size_t)i * c.max_pos * d;
    Linear lp; lp.w = L.pos_w; lp.b = nullptr; lp.out = d; lp.in = d;
    WN_TRY(linear(lp, m->pe, d, L.pos_tab, d, c.max_pos, 0));
  }
  auto DEC = [&](Decoder& D, const std::string& pfx, int nl) {
    D.embed = W(pfx + ".embed");
    D.pe = m->pe;  // same sinusoid table (embedding.py:47-56), same d_model
    D.after = NORM(pfx + ".after_norm");
    D.out = LIN(pfx + ".output_layer", V, d);
    D.layers.resize(nl);
    for (int j = 0; j < nl; ++j) {
      const std::string p = pfx + ".decoders." + std::to_string(j);
      DecLayer& L = D.layers[j];
      L.n1 = NORM(p + ".norm1"); L.n2 = NORM(p + ".norm2"); L.n3 = NORM(p + ".norm3");
      L.self_qkv = LIN(p + ".self_qkv", 3 * d, d);
      L.self_out = LIN(p + ".self_attn.linear_out", d, d);
      L.src_q = LIN(p + ".src_attn.linear_q", d, d);
      L.src_kv = LIN(p + ".src_kv", 2 * d, d);
      L.src_out = LIN(p + ".src_attn.linear_out", d, d);
      L.ff1 = LIN(p + ".feed_forward.w_1", c.dec_ffn_dim, d);
      L.ff2 = LIN(p + ".feed_forward.w_2", d, c.dec_ffn_dim);
    }
  };
  if (has_dec) {
    if (c.bidirectional) {
      DEC(m->left, "decoder.left_decoder", c.dec_layers);
      if (c.dec_r_layers > 0) DEC(m->right, "decoder.right_decoder", c.dec_r_layers);
    } else {
      DEC(m->left, "decoder", c.dec_layers);
    }
  }
  WN_TRY(build_x6_images(m.get()));
  WN_HIP(hipDeviceSynchronize());
  *out = m.release();
  return 0;
}

void wn_model_destroy(wn_model* m) { delete m; }

int wn_model_clone(const wn_model* src, wn_model** out) {
  WN_CHECK(src && out, "wn_model_clone: null argument");
  WN_HIP(hipSetDevice(src->device));
  std::unique_ptr<wn_model> m(new wn_model());
  m->cfg = src->cfg;
  m->device = src->device;
  m->prec = src->prec;
  // weights, projected position tables and fbank tables are read-only: share
  m->weights = src->weights;
  m->n_weight_elems = src->n_weight_elems;
  m->weights_bf16 = src->weights_bf16;
  m->weights_mx = src->weights_mx; m->mx_at = src->mx_at; m->fp8_ffn = src->fp8_ffn;
  m->weights_x6 = src->weights_x6; m->x6_at = src->x6_at;
  m->weights_x6p = src->weights_x6p; m->x6p_at = src->x6p_at; m->bias4_buf = src->bias4_buf; m->bias4 = src->bias4;
  m->pos_tabs = src->pos_tabs;
  m->fb_tab_i = src->fb_tab_i;
  m->w = src->w;
  m->conv1_w = src->conv1_w; m->conv1_b = src->conv1_b;
  m->conv2 = src->conv2; m->sub_out = src->sub_out;
  m->cmvn_mean = src->cmvn_mean; m->cmvn_istd = src->cmvn_istd;
  m->pe = src->pe;
  m->after_norm = src->after_norm;
  m->ctc = src->ctc;
  m->layers = src->layers;
  m->tf_layers = src->tf_layers;
  m->tconv1 = src->tconv1; m->tconv2 = src->tconv2;
  m->fbank_ok = src->fbank_ok;
  m->lm_dft = src->lm_dft; m->lm_mel = src->lm_mel;
  m->rs_taps = src->rs_taps;
  m->left = src->left; m->right = src->right;
  m->fb_window = src->fb_window; m->fb_twiddle = src->fb_twiddle;
  m->fb_mel_w = src->fb_mel_w;
  m->ctx_buf = src->ctx_buf; m->ctx = src->ctx;
  *out = m.release();
  return 0;
}

int wn_model_set_precision(wn_model* m, int32_t precision) {
  WN_CHECK(m, "wn_model_set_precision: null model");
  WN_CHECK(precision == PREC_F32 || precision == PREC_BF16 || precision == PREC_FP8,
           "wn_model_set_precision: 0 (fp32), 1 (bf16 operands, fp32 accumulate) or 2 "
           "(bf16 + MXFP8 feed-forward GEMMs)");
  if (precision != PREC_F32 && !m->weights_bf16 && m->n_weight_elems > 0) {
    // one-time bf16 image of the weight slab for the bf16-storage GEMMs (same
    // element offsets; clones made afterwards share it)
    WN_HIP(hipSetDevice(m->device));
    auto img = std::make_shared<DevBuf>();
    WN_TRY(img->ensure((size_t)m->n_weight_elems * 2));
    WN_TRY(convert_f32_to_bf16(m->weights->as<float>(), img->p, m->n_weight_elems,
                               nullptr));
    WN_HIP(hipStreamSynchronize(nullptr));
    m->weights_bf16 = img;
  }
  if (precision == PREC_FP8 && !m->mx_at) {
    // one-time MXFP8 images of the feed-forward weights (w_1, w_2 of every encoder
    // layer): e4m3 [N][K] + block scales [K/128][N]
    WN_HIP(hipSetDevice(m->device));
    std::vector<const Linear*> ws;
    for (const auto& L : m->layers) { ws.push_back(&L.ffm1); ws.push_back(&L.ffm2);
                                      ws.push_back(&L.ff1); ws.push_back(&L.ff2); }
    for (const auto& L : m->tf_layers) { ws.push_back(&L.ff1); ws.push_back(&L.ff2); }
    size_t bytes = 0;
    for (const Linear* l : ws)
      if (l->w && l->in % 128 == 0)
        bytes += ((size_t)l->out * l->in + 255) / 256 * 256 + (size_t)(l->in / 128) * l->out * 4;
    auto buf = std::make_shared<DevBuf>();
    auto at = std::make_shared<std::map<const float*, wn_model::MxW>>();
    if (bytes > 0) {
      WN_TRY(buf->ensure(bytes));
      char* p = buf->as<char>();
      for (const Linear* l : ws) {
        if (!l->w || l->in % 128 != 0) continue;
        char* q = p;
        p += ((size_t)l->out * l->in + 255) / 256 * 256;
        unsigned* sc = reinterpret_cast<unsigned*>(p);
        p += (size_t)(l->in / 128) * l->out * 4;
        WN_TRY(mx_quantize(l->w, l->in, l->out, l->in, q, sc, l->out, nullptr));
        (*at)[l->w] = wn_model::MxW{q, sc};
      }
      WN_HIP(hipStreamSynchronize(nullptr));
    }
    m->weights_mx = buf;
    m->mx_at = at;
  }
  m->prec = precision == PREC_F32 ? PREC_F32 : PREC_BF16;
  m->fp8_ffn = precision == PREC_FP8;
  return 0;
}

int32_t wn_model_get_precision(const wn_model* m) {
  return m ? (m->fp8_ffn ? (int32_t)PREC_FP8 : m->prec) : -1;
}

int wn_profile_enable(wn_model* m, int32_t on) {
  WN_CHECK(m, "wn_profile_enable: null model");
  m->prof_on = on != 0;
  m->prof_used = 0;
  m->prof_flops = 0.0;
  return 0;
}

const char* wn_profile_kernel_name(const wn_model* m) {
  return m ? m->prof_kernel : "";
}

int32_t wn_profile_ffn_split(const wn_model* m) { return m ? m->prof_split : 0; }

int wn_profile_collect(wn_model* m, int32_t* n_launches, double* total_ms,
                       double* total_flops) {
  WN_CHECK(m && n_launches && total_ms && total_flops, "wn_profile_collect: null");
  double ms = 0.0;
  for (size_t i = 0; i + 1 < m->prof_used; i += 2) {
    WN_HIP(hipEventSynchronize(m->prof_ev[i + 1]));
    float t = 0.f;
    WN_HIP(hipEventElapsedTime(&t, m->prof_ev[i], m->prof_ev[i + 1]));
    ms += t;
  }
  *n_launches = (int32_t)(m->prof_used / 2);
  *total_ms = ms;
  *total_flops = m->prof_flops;
  m->prof_used = 0;
  m->prof_flops = 0.0;
  return 0;
}

int wn_debug_set(wn_model* m, const char* key, int32_t value) {
  WN_CHECK(m && key, "wn_debug_set: null argument");
  const std::string k(key);
  if (k == "n_layers") m->dbg_layers = value;
  else if (k == "skip_after_norm") m->dbg_skip_after_norm = value;
  else { set_error("wn_debug_set: unknown key " + k); return -1; }
  return 0;
}

int wn_tune_set(const char* key, int32_t value) {
  WN_CHECK(key, "wn_tune_set: null key");
  const std::string k(key);
  if (k == "gemm_variant") g_gemm_variant = value;
  else if (k == "gemm_tile") g_gemm_tile = value;
  else if (k == "gemm_tile_conv") g_gemm_tile_conv = value;
  else if (k == "gemm_tile_glu") g_gemm_tile_glu = value;
  else if (k == "gemm_tile_bf16") g_gemm_tile_bf16 = value;
  else if (k == "ln_rows") g_ln_rows = value;
  else if (k == "attn_split") g_attn_split = value;
  else if (k == "attn_bf16") g_attn_bf16 = value;
  else if (k == "bf16_store") g_bf16_store = value;
  else if (k == "attn_bf16_nw") g_attn_bf16_nw = value;
  else if (k == "attn_bf16_sub") g_attn_bf16_sub = value;
  else if (k == "qkv_bf16") g_qkv_bf16 = value;
  else if (k == "fp8_min_tiles") g_fp8_min_tiles = value;
  else if (k == "ffn_fused") g_ffn_fused = value;
  else if (k == "ffn_bm64") g_ffn_bm64 = value;
  else if (k == "gemm_x6") g_gemm_x6 = value;
  else if (k == "x6_conv_bm") g_x6_conv_bm = value;
  else if (k == "x6_ffn_s") g_x6_ffn_s = value;
  else if (k == "x6_probe") g_x6_probe = value;
  else if (k == "x6_nw4") g_x6_nw4 = value;
  else if (k == "x6_conv") g_x6_conv = value;
  else if (k == "attn_fold") g_attn_fold = value;
  else if (k == "x6_conv_order") g_x6_conv_order = value;
  else if (k == "x6_linear") g_x6_linear = value;
  else if (k == "x6_linear_min") g_x6_linear_min = value;
  else if (k == "x6_af32") g_x6_af32 = value;
  else if (k == "beam_prio") g_beam_prio = value;
  else if (k == "beam_weak_hash") g_beam_weak_hash = value;
  else if (k == "ctc_wave") g_ctc_wave = value;
  else if (k == "gemm_rowln") g_gemm_rowln = value;
  else if (k == "x6r") g_x6r = value;
  else if (k == "ffn_ring") g_ffn_ring = value;
  else if (k == "ffn_x6f") g_ffn_x6f = value;
  else if (k == "ffn_x6f_ring") g_ffn_x6f_ring = value;
  else if (k == "ffn_x6f_var") g_ffn_x6f_var = value;
  else { set_error("wn_tune_set: unknown key " + k); return -1; }
  return 0;
}

int wn_workspace_create(int32_t device, wn_model** out) {
  WN_CHECK(out, "wn_workspace_create: null argument");
  WN_HIP(hipSetDevice(device));
  wn_model* m = new wn_model();
  memset(&m->cfg, 0, sizeof(m->cfg));
  m->device = device;
  *out = m;
  return 0;
}

// ---------------------------------------------------------------------------
int wn_encode(wn_model* m, const float* feats_dev, const int32_t* feat_lens_host,
              int32_t B, int32_t T, int32_t chunk, int32_t left,
              float* enc_out_dev, int32_t* enc_lens_host, void* stream) {
  WN_CHECK(m && feats_dev && feat_lens_host, "wn_encode: null argument");
  WN_ENTER(m);
  PrecisionScope prec_scope(m);
  WN_CHECK(!m->layers.empty() || !m->tf_layers.empty(),
           "wn_encode: this handle has no weights");
  WN_CHECK(B > 0, "wn_encode: empty batch");
  WN_CHECK(chunk != 0, "decoding_chunk_size must not be 0 (asr_model.py:310)");
  if (m->cfg.encoder_type == 1) {
    WN_CHECK(chunk < 0 && m->cfg.static_chunk_size <= 0,
             "chunk decoding is not implemented for the transformer encoder");
    WN_CHECK(T >= 1, "wn_encode: empty features");
    WN_HIP(hipSetDevice(m->device));
    return encode_transformer(m, feats_dev, feat_lens_host, B, T, enc_out_dev,
                              enc_lens_host, (hipStream_t)stream);
  }
  WN_CHECK(T >= 7, "wn_encode: at least 7 frames are needed by Conv2dSubsampling4");
  hipStream_t s = (hipStream_t)stream;
  WN_HIP(hipSetDevice(m->device));
  const wn_config& c = m->cfg;
  const int d = c.d_model;
  const int Tp = ((T - 1) / 2 - 1) / 2;
  WN_TRY(subsample_conv2d4(m, feats_dev, feat_lens_host, B, T, enc_lens_host, 0, s));
  const int M = m->rows;
  if (M > 0) {
    WN_TRY(encoder_layers(m, chunk, left, s));
  }
  if (enc_out_dev) {
    if (M > 0) {
      hipLaunchKernelGGL(scatter_padded_kernel, dim3(Tp, B), dim3(64), 0, s,
                         m->enc.as<float>(), d, m->d_off.as<int>(),
                         m->d_len.as<int>(), Tp, d / 4, enc_out_dev);
      WN_HIP(hipGetLastError());
    } else if (Tp > 0) {
      WN_HIP(hipMemsetAsync(enc_out_dev, 0, (size_t)B * Tp * d * sizeof(float), s));
    }
  }
  return 0;
}

int wn_encode_chunk_batch(wn_model* m, int32_t n_sess, const float* feats_dev, int32_t time,
                          const int32_t* offsets_host, int32_t required_cache_size,
                          const float* const* att_cache_dev, const int32_t* cache_t1_host,
                          const float* const* cnn_cache_dev, float* out_dev,
                          float* const* new_att_cache_dev, float* const* new_cnn_cache_dev,
                          int32_t* chunk_out, int32_t* new_cache_t1_out, void* stream) {
  WN_CHECK(m && feats_dev && out_dev && offsets_host && cache_t1_host && n_sess >= 1,
           "wn_encode_chunk: null argument");
  WN_ENTER(m);
  PrecisionScope prec_scope(m);
  WN_CHECK(!m->layers.empty() && m->cfg.encoder_type == 0,
           "wn_encode_chunk: needs a Conformer encoder");
  WN_CHECK(time >= 7, "wn_encode_chunk: at least 7 frames are needed by Conv2dSubsampling4");
  hipStream_t s = (hipStream_t)stream;
  WN_HIP(hipSetDevice(m->device));
  const wn_config& c = m->cfg;
  const int R = ((time - 1) / 2 - 1) / 2;
  const int lorder = c.causal ? c.cnn_kernel - 1 : 0;
  std::vector<ChunkSess> sess(n_sess);
  std::vector<int32_t> lens(n_sess, time);
  for (int b = 0; b < n_sess; ++b) {
    const int offset = offsets_host[b], t1c = cache_t1_host[b];
    WN_CHECK(offset >= 0 && t1c >= 0 && t1c <= offset,
             "wn_encode_chunk: need 0 <= cache_t1 <= offset");
    WN_CHECK(t1c == 0 || (att_cache_dev && att_cache_dev[b]), "wn_encode_chunk: att_cache is null");
    WN_CHECK(offset + R <= c.max_pos, "wn_encode_chunk: offset beyond the positional table");
    const int key = t1c + R;
    // encoder.py:258-263
    const int next_start = required_cache_size < 0 ? 0
                           : required_cache_size == 0 ? key
                           : std::max(key - required_cache_size, 0);
    const int nt = key - next_start;
    WN_CHECK(nt == 0 || (new_att_cache_dev && new_att_cache_dev[b]),
             "wn_encode_chunk: new_att_cache is null");
    WN_CHECK(lorder == 0 || (new_cnn_cache_dev && new_cnn_cache_dev[b]),
             "wn_encode_chunk: new_cnn_cache is null");
    ChunkSess& ss = sess[b];
    ss.att_cache = t1c > 0 ? att_cache_dev[b] : nullptr;
    // nt == 0: the kernel writes no cache rows, any non-null pointer will do
    ss.new_att = nt > 0 ? new_att_cache_dev[b] : out_dev;
    ss.cnn_cache = (cnn_cache_dev && lorder > 0) ? cnn_cache_dev[b] : nullptr;
    ss.new_cnn = lorder > 0 ? new_cnn_cache_dev[b] : nullptr;
    ss.t1 = t1c; ss.next_start = next_start; ss.nt = nt; ss.kv_off = 0;
    if (new_cache_t1_out) new_cache_t1_out[b] = nt;
  }
  WN_TRY(subsample_conv2d4(m, feats_dev, lens.data(), n_sess, time, nullptr,
                           offsets_host[0], s));
  WN_CHECK(m->rows == n_sess * R, "wn_encode_chunk: internal row count");
  WN_TRY(encoder_layers_chunk(m, n_sess, R, offsets_host, sess, out_dev, s));
  m->rows = 0; m->B = 0;  // the handle holds no decodable batch after a chunk call
  if (chunk_out) *chunk_out = R;
  return 0;
}

int wn_encode_chunk(wn_model* m, const float* feats_dev, int32_t time, int32_t offset,
                    int32_t required_cache_size, const float* att_cache_dev,
                    int32_t cache_t1, const float* cnn_cache_dev, float* out_dev,
                    float* new_att_cache_dev, float* new_cnn_cache_dev,
                    int32_t* chunk_out, int32_t* new_cache_t1_out, void* stream) {
  return wn_encode_chunk_batch(m, 1, feats_dev, time, &offset, required_cache_size,
                               &att_cache_dev, &cache_t1, &cnn_cache_dev, out_dev,
                               &new_att_cache_dev, &new_cnn_cache_dev, chunk_out,
                               new_cache_t1_out, stream);
}

int wn_set_encoder_out(wn_model* m, const float* enc_out_dev,
                       const int32_t* enc_lens_host, int32_t B, int32_t Tp,
                       void* stream) {
  WN_CHECK(m && enc_out_dev && enc_lens_host && B > 0 && Tp > 0,
           "wn_set_encoder_out: bad argument");
  WN_ENTER(m);
  hipStream_t s = (hipStream_t)stream;
  WN_HIP(hipSetDevice(m->device));
  std::vector<int> off(B), len(B);
  for (int b = 0; b < B; ++b) {
    WN_CHECK(enc_lens_host[b] >= 0 && enc_lens_host[b] <= Tp, "length > Tp");
    off[b] = b * Tp; len[b] = enc_lens_host[b];
  }
  WN_TRY(set_layout(m, B, Tp, off, len, B * Tp, s));
  WN_TRY(m->stage.end(s));
  const size_t bytes = (size_t)B * Tp * m->cfg.d_model * sizeof(float);
  WN_TRY(m->enc.ensure(bytes));
  WN_HIP(hipMemcpyAsync(m->enc.p, enc_out_dev, bytes, hipMemcpyDeviceToDevice, s));
  return 0;
}

// ---------------------------------------------------------------------------
int wn_ctc_logprobs(wn_model* m, int32_t topk, int32_t blank_id,
                    float blank_penalty, float* logp_dev, int32_t Tp,
                    void* stream) {
  WN_CHECK(m && m->B > 0, "wn_ctc_logprobs: no current batch (call wn_encode)");
  WN_ENTER(m);
  PrecisionScope prec_scope(m);
  WN_CHECK(m->ctc.w, "wn_ctc_logprobs: this handle has no weights");
  hipStream_t s = (hipStream_t)stream;
  WN_HIP(hipSetDevice(m->device));
  const wn_config& c = m->cfg;
  const int M = m->rows, V = c.vocab;
  const int k = std::max(1, topk);
  WN_CHECK(k <= V, "top-k larger than the vocabulary");
  WN_CHECK(!logp_dev || Tp == m->Tp, "wn_ctc_logprobs: Tp mismatch");
  m->ctc_rows = M; m->ctc_k = k;
  int ldv = V;
  if (M > 0) {
    // logits rows at a pitch of V rounded up to 4 floats (16-B aligned rows; what the
    // six-product GEMM needs to store them)
    const int V4 = (V + 3) / 4 * 4;
    ldv = V4;
    WN_TRY(m->logits.ensure((size_t)M * V4 * sizeof(float)));
    WN_TRY(m->topk_val.ensure((size_t)M * k * sizeof(float)));
    WN_TRY(m->topk_idx.ensure((size_t)M * k * sizeof(int)));
    WN_TRY(vocab_linear(m, m->ctc, m->enc.as<float>(), c.d_model, m->logits.as<float>(), M, s));
    CtcRowArgs r;
    r.logits = m->logits.as<float>(); r.ld = V4; r.M = M; r.V = V; r.k = k;
    r.blank = blank_id; r.blank_penalty = blank_penalty > 0.f ? blank_penalty : 0.f;
    r.topk_val = m->topk_val.as<float>(); r.topk_idx = m->topk_idx.as<int>();
    // normalised rows are written back in place when the caller wants them
    r.logp = logp_dev ? m->logits.as<float>() : nullptr; r.ld_out = V4;
    WN_TRY(ctc_logsoftmax_topk(r, s));
  }
  if (logp_dev) {
    if (M > 0) {
      hipLaunchKernelGGL(scatter_padded_any_kernel, dim3(m->Tp, m->B), dim3(256),
                         0, s, m->logits.as<float>(), ldv, m->d_off.as<int>(),
                         m->d_len.as<int>(), m->Tp, V, logp_dev);
      WN_HIP(hipGetLastError());
    } else {
      WN_HIP(hipMemsetAsync(logp_dev, 0, (size_t)m->B * m->Tp * V * sizeof(float), s));
    }
  }
  m->ctc_valid = true;
  return 0;
}

namespace {
__global__ __launch_bounds__(256) void topk_raw_kernel(const float* x, int ld,
                                                        int V, int k,
                                                        float* tv, int* ti) {
  // top-k of an already normalised row (no log-softmax): k block-argmax rounds
  __shared__ float rv[4];
  __shared__ int ri[4];
  __shared__ float cv;
  __shared__ int ci;
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* p = x + (int64_t)row * ld;
  float pv = INFINITY;
  int pi = -1;
  for (int r = 0; r < k; ++r) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < V; i += 256) {
      const float v = p[i];
      if ((v < pv || (v == pv && i > pi)) && (v > bv || (v == bv && i < bi))) {
        bv = v; bi = i;
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { rv[wave] = bv; ri[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 4; ++w)
        if (rv[w] > rv[0] || (rv[w] == rv[0] && ri[w] < ri[0])) { rv[0] = rv[w]; ri[0] = ri[w]; }
      cv = rv[0]; ci = ri[0];
      tv[(int64_t)row * k + r] = cv;
      ti[(int64_t)row * k + r] = ci;
    }
    __syncthreads();
    pv = cv; pi = ci;
  }
}
}  // namespace

int wn_set_ctc_probs(wn_model* m, const float* logp_dev, const int32_t* lens_host,
                     int32_t B, int32_t Tp, int32_t V, int32_t topk,
                     void* stream) {
  WN_CHECK(m && logp_dev && lens_host && B > 0 && Tp > 0 && V > 0,
           "wn_set_ctc_probs: bad argument");
  WN_ENTER(m);
  hipStream_t s = (hipStream_t)stream;
  WN_HIP(hipSetDevice(m->device));
  const int k = std::max(1, topk);
  WN_CHECK(k <= V, "top-k larger than the vocabulary");
  std::vector<int> off(B), len(B);
  for (int b = 0; b < B; ++b) {
    WN_CHECK(lens_host[b] >= 0 && lens_host[b] <= Tp, "length > Tp");
    off[b] = b * Tp; len[b] = lens_host[b];
  }
  WN_TRY(set_layout(m, B, Tp, off, len, B * Tp, s));
  WN_TRY(m->stage.end(s));
  const int M = B * Tp;
  WN_TRY(m->topk_val.ensure((size_t)M * k * sizeof(float)));
  WN_TRY(m->topk_idx.ensure((size_t)M * k * sizeof(int)));
  hipLaunchKernelGGL(topk_raw_kernel, dim3(M), dim3(256), 0, s, logp_dev, V, V,
                     k, m->topk_val.as<float>(), m->topk_idx.as<int>());
  WN_HIP(hipGetLastError());
  m->ctc_rows = M; m->ctc_k = k; m->ctc_valid = true;
  return 0;
}

int wn_ctc_greedy_search(wn_model* m, int32_t blank_id, int32_t* tokens_host,
                         int32_t* tok_lens_host, int32_t max_len, void* stream) {
  WN_CHECK(m && m->ctc_valid, "greedy: no CTC posteriors (call wn_ctc_logprobs)");
  WN_ENTER(m);
  WN_CHECK(tokens_host && tok_lens_host, "greedy: null output");
  hipStream_t s = (hipStream_t)stream;
  WN_HIP(hipSetDevice(m->device));
  const int B = m->B;
  int longest = 0;
  for (int b = 0; b < B; ++b) longest = std::max(longest, m->len[b]);
  WN_CHECK(max_len >= longest, "greedy: max_len smaller than the longest utterance");
  const int ml = std::max(max_len, 1);
  WN_TRY(m->g_tok.ensure((size_t)B * ml * sizeof(int)));
  WN_TRY(m->g_len.ensure((size_t)B * sizeof(int)));
  WN_TRY(ctc_greedy_collapse(m->topk_idx.as<int>(), m->ctc_k, m->d_off.as<int>(),
                             m->d_len.as<int>(), B, blank_id, m->g_tok.as<int>(),
                             ml, m->g_len.as<int>(), s));
  WN_HIP(hipMemcpyAsync(tokens_host, m->g_tok.p, (size_t)B * ml * sizeof(int),
                        hipMemcpyDeviceToHost, s));
  WN_HIP(hipMemcpyAsync(tok_lens_host, m->g_len.p, (size_t)B * sizeof(int),
                        hipMemcpyDeviceToHost, s));
  WN_HIP(hipStreamSynchronize(s));
  return 0;
}

int wn_filter_blank_embedding(wn_model* m, float* padded_out_dev, int32_t* n_keep_host,
                              int32_t* t_out, void* stream) {
  WN_CHECK(m && m->B > 0 && m->enc.p && m->ctc_valid && n_keep_host && t_out,
           "filter_blank_embedding: needs the encoder output and the CTC posteriors of the "
           "current batch (wn_encode / wn_set_encoder_out, then wn_ctc_logprobs)");
  WN_ENTER(m);
  hipStream_t s = (hipStream_t)stream;
  WN_HIP(hipSetDevice(m->device));
  const int B = m->B, d = m->cfg.d_model, M = m->rows;
  WN_CHECK(m->ctc_rows == M, "filter_blank_embedding: CTC posteriors of another layout");
  WN_TRY(m->nb_map.ensure((size_t)std::max(M, 1) * sizeof(int)));
  WN_TRY(m->nb_keep.ensure((size_t)B * sizeof(int)));
  WN_TRY(nonblank_map(m->topk_idx.as<int>(), m->ctc_k, m->d_off.as<int>(), m->d_len.as<int>(), B,
                      m->nb_map.as<int>(), m->nb_keep.as<int>(), s));
  std::vector<int> keep(B);
  WN_HIP(hipMemcpyAsync(keep.data(), m->nb_keep.p, (size_t)B * sizeof(int),
                        hipMemcpyDeviceToHost, s));
  WN_HIP(hipStreamSynchronize(s));
  int T = 0;
  for (int b = 0; b < B; ++b) { n_keep_host[b] = keep[b]; T = std::max(T, keep[b]); }
  *t_out = T;
  WN_CHECK(T > 0, "filter_blank_embedding: no non-blank frame in the whole batch");
  // new layout: utterance b keeps min(len[b], T) rows -- attention_rescoring slices the
  // zero-padded (B, T, d) tensor with the UNFILTERED lengths (asr_model.py:337-342,
  // search.py:396): the selected rows, then zero rows the decoder attends to as well
  std::vector<int> noff(B), nlen(B), old_off = m->off;
  int rows = 0;
  for (int b = 0; b < B; ++b) { noff[b] = rows; nlen[b] = std::min(m->len[b], T); rows += nlen[b]; }
  WN_TRY(m->nb_enc.ensure((size_t)std::max(rows, 1) * d * sizeof(float)));
  // descriptors of the OLD layout stay valid on the device until set_layout replaces them:
  // gather first (it reads d_off of the old layout through a private copy)
  WN_TRY(m->nb_off_old.ensure((size_t)B * sizeof(int)));
  WN_HIP(hipMemcpyAsync(m->nb_off_old.p, m->d_off.p, (size_t)B * sizeof(int),
                        hipMemcpyDeviceToDevice, s));
  WN_TRY(set_layout(m, B, T, noff, nlen, rows, s));
  WN_TRY(m->stage.end(s));
  WN_TRY(nonblank_gather(m->enc.as<float>(), m->nb_map.as<int>(), m->nb_off_old.as<int>(),
                         m->nb_keep.as<int>(), m->d_off.as<int>(), m->d_len.as<int>(),
                         m->d_row_utt.as<int>(), m->nb_enc.as<float>(), d, rows, s));
  std::swap(m->enc.p, m->nb_enc.p);
  std::swap(m->enc.cap, m->nb_enc.cap);
  if (padded_out_dev) {
    // the reference's return value: (B, T, d), utterance b's selected rows then zeros
    WN_HIP(hipMemsetAsync(padded_out_dev, 0, (size_t)B * T * d * sizeof(float), s));
    for (int b = 0; b < B; ++b)
      if (keep[b] > 0)
        WN_HIP(hipMemcpyAsync(padded_out_dev + (size_t)b * T * d,
                              m->enc.as<float>() + (size_t)noff[b] * d,
                              (size_t)std::min(keep[b], nlen[b]) * d * sizeof(float),
                              hipMemcpyDeviceToDevice, s));
  }
  return 0;
}

int wn_set_context_graph(wn_model* m, int32_t n_nodes, const int32_t* fail,
                         const double* node_score, const double* output_score,
                         const double* token_score, int32_t n_edges,
                         const int32_t* edge_from, const int32_t* edge_token,
                         const int32_t* edge_to, void* stream) {
  WN_CHECK(m, "context graph: null model");
  WN_ENTER(m);
  if (n_nodes <= 0) {
    m->ctx = CtxGraph();
    m->ctx_buf.reset();
    return 0;
  }
  WN_CHECK(fail && node_score && output_score && token_score,
           "context graph: null node array");
  WN_CHECK(n_edges >= 0 && (n_edges == 0 || (edge_from && edge_token && edge_to)),
           "context graph: null edge array");
  WN_CHECK(fail[0] == 0, "context graph: node 0 must be the root (fail[0] == 0)");
  for (int i = 0; i < n_nodes; ++i)
    WN_CHECK(fail[i] >= 0 && fail[i] < n_nodes, "context graph: fail arc out of range");
  hipStream_t s = (hipStream_t)stream;
  WN_HIP(hipSetDevice(m->device));
  unsigned slots = 16;
  while (slots < 2u * (unsigned)n_edges) slots *= 2;
  std::vector<unsigned long long> keys(slots, CTX_EMPTY);
  std::vector<int> vals(slots, -1);
  for (int i = 0; i < n_edges; ++i) {
    WN_CHECK(edge_from[i] >= 0 && edge_from[i] < n_nodes && edge_to[i] > 0 &&
                 edge_to[i] < n_nodes && edge_token[i] >= 0,
             "context graph: edge out of range");
    const unsigned long long key =
        ((unsigned long long)(unsigned)edge_from[i] << 32) | (unsigned)edge_token[i];
    unsigned h = ctx_slot(key, slots - 1);
    while (keys[h] != CTX_EMPTY) {
      WN_CHECK(keys[h] != key, "context graph: duplicate edge");
      h = (h + 1) & (slots - 1);
    }
    keys[h] = key;
    vals[h] = edge_to[i];
  }
  // one slab: keys | 3 x double[n] | fail[n] | vals[slots]
  const size_t o_keys = 0;
  const size_t o_ns = o_keys + slots * sizeof(unsigned long long);
  const size_t o_os = o_ns + (size_t)n_nodes * sizeof(double);
  const size_t o_ts = o_os + (size_t)n_nodes * sizeof(double);
  const size_t o_fail = o_ts + (size_t)n_nodes * sizeof(double);
  const size_t o_vals = o_fail + (size_t)n_nodes * sizeof(int);
  const size_t total = o_vals + slots * sizeof(int);
  std::vector<char> host(total);
  memcpy(host.data() + o_keys, keys.data(), slots * sizeof(unsigned long long));
  memcpy(host.data() + o_ns, node_score, (size_t)n_nodes * sizeof(double));
  memcpy(host.data() + o_os, output_score, (size_t)n_nodes * sizeof(double));
  memcpy(host.data() + o_ts, token_score, (size_t)n_nodes * sizeof(double));
  memcpy(host.data() + o_fail, fail, (size_t)n_nodes * sizeof(int));
  memcpy(host.data() + o_vals, vals.data(), slots * sizeof(int));
  // a fresh buffer: clones of this handle may still search with the old one
  auto buf = std::make_shared<DevBuf>();
  WN_TRY(buf->ensure(total));
  WN_HIP(hipMemcpyAsync(buf->p, host.data(), total, hipMemcpyHostToDevice, s));
  WN_HIP(hipStreamSynchronize(s));
  char* base = buf->as<char>();
  CtxGraph g;
  g.keys = reinterpret_cast<const unsigned long long*>(base + o_keys);
  g.node_score = reinterpret_cast<const double*>(base + o_ns);
  g.output_score = reinterpret_cast<const double*>(base + o_os);
  g.token_score = reinterpret_cast<const double*>(base + o_ts);
  g.fail = reinterpret_cast<const int*>(base + o_fail);
  g.vals = reinterpret_cast<const int*>(base + o_vals);
  g.mask = slots - 1;
  m->ctx_buf = buf;
  m->ctx = g;
  return 0;
}

int wn_ctc_prefix_beam_search(wn_model* m, int32_t beam, int32_t blank_id,
                              int32_t* n_hyps_host, int32_t* hyp_lens_host,
                              int32_t* hyp_tlens_host, int32_t* hyp_tokens_host,
                              int32_t* hyp_times_host, double* hyp_scores_host,
                              int32_t max_len, void* stream) {
  WN_CHECK(m && m->ctc_valid, "prefix beam: no CTC posteriors");
  WN_ENTER(m);
  WN_CHECK(m->ctc_k == beam, "prefix beam: wn_ctc_logprobs must be called with topk == beam");
  WN_CHECK(n_hyps_host && hyp_lens_host && hyp_tlens_host && hyp_tokens_host &&
               hyp_times_host && hyp_scores_host, "prefix beam: null output");
  hipStream_t s = (hipStream_t)stream;
  WN_HIP(hipSetDevice(m->device));
  const int B = m->B;
  int longest = 0;
  for (int b = 0; b < B; ++b) longest = std::max(longest, m->len[b]);
  WN_CHECK(max_len >= longest && max_len >= 1, "prefix beam: max_len too small");
  const int64_t pool = prefix_beam_pool_ints(max_len, beam);
  WN_TRY(m->pb_pool.ensure((size_t)B * pool * sizeof(int)));
  const size_t nb = (size_t)B * beam;
  WN_TRY(m->pb_nh.ensure(B * sizeof(int)));
  WN_TRY(m->pb_len.ensure(nb * sizeof(int)));
  WN_TRY(m->pb_tlen.ensure(nb * sizeof(int)));
  WN_TRY(m->pb_tok.ensure(nb * max_len * sizeof(int)));
  WN_TRY(m->pb_tim.ensure(nb * max_len * sizeof(int)));
  WN_TRY(m->pb_score.ensure(nb * sizeof(double)));
  PrefixBeamArgs a;
  a.topk_val = m->topk_val.as<float>(); a.topk_idx = m->topk_idx.as<int>();
  a.k = m->ctc_k; a.off = m->d_off.as<int>(); a.len = m->d_len.as<int>();
  a.B = B; a.beam = beam; a.blank = blank_id; a.max_len = max_len;
  a.pool = m->pb_pool.as<int>(); a.pool_stride = pool;
  a.n_hyps = m->pb_nh.as<int>(); a.hyp_lens = m->pb_len.as<int>();
  a.hyp_tlens = m->pb_tlen.as<int>(); a.hyp_tokens = m->pb_tok.as<int>();
  a.hyp_times = m->pb_tim.as<int>(); a.hyp_scores = m->pb_score.as<double>();
  a.cg = m->ctx;
  static const bool pb_dbg = getenv("WN_PB_CYCLES") != nullptr;  // debugging aid
  if (pb_dbg) {
    WN_TRY(m->pb_dbg.ensure(8 * sizeof(long long)));
    a.dbg_cycles = m->pb_dbg.as<long long>();
  }
  WN_TRY(ctc_prefix_beam(a, s));
  if (pb_dbg) {
    long long h[5];
    WN_HIP(hipMemcpyAsync(h, a.dbg_cycles, sizeof(h), hipMemcpyDeviceToHost, s));
    WN_HIP(hipStreamSynchronize(s));
    fprintf(stderr, "[wn] prefix beam wg0: frames %lld, cycles/frame eval %.0f rank %.0f "
            "select %.0f; emit %lld cycles\n", h[3], (double)h[0] / h[3],
            (double)h[1] / h[3], (double)h[2] / h[3], h[4]);
  }
  WN_HIP(hipMemcpyAsync(n_hyps_host, a.n_hyps, B * sizeof(int), hipMemcpyDeviceToHost, s));
  WN_HIP(hipMemcpyAsync(hyp_lens_host, a.hyp_lens, nb * sizeof(int), hipMemcpyDeviceToHost, s));
  WN_HIP(hipMemcpyAsync(hyp_tlens_host, a.hyp_tlens, nb * sizeof(int), hipMemcpyDeviceToHost, s));
  WN_HIP(hipMemcpyAsync(hyp_tokens_host, a.hyp_tokens, nb * max_len * sizeof(int), hipMemcpyDeviceToHost, s));
  WN_HIP(hipMemcpyAsync(hyp_times_host, a.hyp_times, nb * max_len * sizeof(int), hipMemcpyDeviceToHost, s));
  WN_HIP(hipMemcpyAsync(hyp_scores_host, a.hyp_scores, nb * sizeof(double), hipMemcpyDeviceToHost, s));
  WN_HIP(hipStreamSynchronize(s));
  return 0;
}

// ---------------------------------------------------------------------------
namespace {
// embed + the decoder layers over a ragged batch of R token rows (n_seq
// sequences); the result stays in m->r_x.  With `mem_cache` the cross-attention
// K/V projections of the encoder output are computed once per batch and layer
// and reused by later calls (the autoregressive search calls this per step).
int decoder_layers(wn_model* m, const Decoder& D, int R, int n_seq, int max_q,
                   const int* d_tok, bool mem_cache, hipStream_t s,
                   const int* self_kvlen = nullptr) {
  const wn_config& c = m->cfg;
  const int d = c.d_model, Menc = m->rows;
  float* x = m->r_x.as<float>();
  float* t1 = m->r_t1.as<float>();
  float* t2 = m->r_t2.as<float>();
  float* qkv = m->r_qkv.as<float>();
  float* hb = m->r_h.as<float>();
  const float eps = c.norm_eps;
  const size_t mem_layer = (size_t)Menc * 2 * d;
  const bool fill_cache = mem_cache && !m->mem_cache_valid;
  if (mem_cache)
    WN_TRY(m->r_mem_all.ensure(D.layers.size() * mem_layer * sizeof(float)));
  // embed(V,d) * sqrt(d) + pe                          embedding.py:58-76
  hipLaunchKernelGGL(embed_kernel, dim3(R), dim3(64), 0, s, d_tok,
                     m->r_pos.as<int>(), D.embed, D.pe, sqrtf((float)d), d / 4, x);
  WN_HIP(hipGetLastError());
  int li = 0;
  for (const DecLayer& L : D.layers) {
    // causal self attention                             decoder_layer.py:100-121
    WN_TRY(ln(L.n1, x, t1, R, d, eps, s));
    WN_TRY(linear(L.self_qkv, t1, d, qkv, 3 * d, R, s));
    AttnArgs a;
    a.Q = qkv; a.K = qkv + d; a.V = qkv + 2 * d; a.ldq = a.ldk = a.ldv = 3 * d;
    a.O = t2; a.ldo = d;
    a.q_off = a.kv_off = m->r_qoff.as<int>();
    a.q_len = a.kv_len = m->r_qlen.as<int>();
    // padded batches (wn_decoder_forward): keys past the sequence length are
    // masked for every query, padded query rows included (mask.py make_pad_mask
    // & subsequent_mask, decoder.py:171-177)
    if (self_kvlen) a.kv_len = self_kvlen;
    a.n_seq = n_seq; a.n_heads = c.dec_heads; a.max_q_len = max_q;
    a.mask_mode = 1; a.scale = 0.125f;
    WN_TRY(attention(a, s));
    WN_TRY(linear(L.self_out, t2, d, x, d, R, s, ACT_NONE, x, d));
    // cross attention over the utterance's encoder frames   decoder_layer.py:123-138
    // (K/V projected once per utterance, not once per hypothesis)
    WN_TRY(ln(L.n2, x, t1, R, d, eps, s));
    WN_TRY(linear(L.src_q, t1, d, t2, d, R, s));
    float* mem = mem_cache ? m->r_mem_all.as<float>() + (size_t)li * mem_layer
                           : m->r_mem.as<float>();
    if (!mem_cache || fill_cache)
      WN_TRY(linear(L.src_kv, m->enc.as<float>(), d, mem, 2 * d, Menc, s));
    AttnArgs cx;
    cx.Q = t2; cx.ldq = d; cx.K = mem; cx.V = mem + d; cx.ldk = cx.ldv = 2 * d;
    cx.O = t1; cx.ldo = d;
    cx.q_off = m->r_qoff.as<int>(); cx.q_len = m->r_qlen.as<int>();
    cx.kv_off = m->r_kvoff.as<int>(); cx.kv_len = m->r_kvlen.as<int>();
    cx.n_seq = n_seq; cx.n_heads = c.dec_heads; cx.max_q_len = max_q;
    cx.mask_mode = 0; cx.scale = 0.125f;
    WN_TRY(attention(cx, s));
    WN_TRY(linear(L.src_out, t1, d, x, d, R, s, ACT_NONE, x, d));
    // FFN (ReLU)                                         decoder_layer.py:140-147
    WN_TRY(ln(L.n3, x, t1, R, d, eps, s));
    WN_TRY(linear(L.ff1, t1, d, hb, c.dec_ffn_dim, R, s, ACT_RELU));
    WN_TRY(linear(L.ff2, hb, c.dec_ffn_dim, x, d, R, s, ACT_NONE, x, d));
    ++li;
  }
  if (fill_cache) m->mem_cache_valid = true;
  return 0;
}

int run_decoder(wn_model* m, const Decoder& D, int R, int n_seq, int max_q,
                const int* d_tok, const int* d_tgt, float* out_dev,
                hipStream_t s) {
  const wn_config& c = m->cfg;
  const int d = c.d_model, V = c.vocab;
  WN_TRY(decoder_layers(m, D, R, n_seq, max_q, d_tok, false, s));
  float* t1 = m->r_t1.as<float>();
  WN_TRY(ln(D.after, m->r_x.as<float>(), t1, R, d, c.norm_eps, s));
  // (the caller sized r_logits for a pitch of V rounded up to 4)
  WN_TRY(vocab_linear(m, D.out, t1, d, m->r_logits.as<float>(), R, s));
  hipLaunchKernelGGL(row_logp_at_kernel, dim3(R), dim3(256), 0, s,
                     m->r_logits.as<float>(), (V + 3) / 4 * 4, V, d_tgt, out_dev);
  WN_HIP(hipGetLastError());
  return 0;
}
}  // namespace

int wn_decoder_next_topk(wn_model* m, int32_t n_seq, const int32_t* seq_utt_host,
                         const int32_t* seq_lens_host, const int32_t* tokens_host,
                         int32_t max_len, int32_t topk, float* logp_host,
                         int32_t* idx_host, void* stream) {
  WN_CHECK(m && m->B > 0 && m->enc.p, "decoder step: no current batch");
  WN_ENTER(m);
  PrecisionScope prec_scope(m);
  WN_CHECK(!m->left.layers.empty(), "decoder step: the model has no attention decoder");
  WN_CHECK(n_seq > 0 && seq_utt_host && seq_lens_host && tokens_host && logp_host &&
               idx_host && max_len > 0, "decoder step: bad argument");
  hipStream_t s = (hipStream_t)stream;
  WN_HIP(hipSetDevice(m->device));
  const wn_config& c = m->cfg;
  const int d = c.d_model, V = c.vocab;
  WN_CHECK(topk >= 1 && topk <= V, "decoder step: top-k");
  std::vector<int> tok, pos, qoff(n_seq), qlen(n_seq), kvoff(n_seq), kvlen(n_seq),
      last(n_seq);
  int max_q = 0;
  for (int i = 0; i < n_seq; ++i) {
    const int u = seq_utt_host[i], L = seq_lens_host[i];
    WN_CHECK(u >= 0 && u < m->B, "decoder step: utterance index");
    WN_CHECK(L >= 1 && L <= max_len && L <= c.max_pos, "decoder step: sequence length");
    WN_CHECK(m->len[u] > 0, "decoder step: utterance without encoder frames");
    qoff[i] = (int)tok.size(); qlen[i] = L;
    kvoff[i] = m->off[u]; kvlen[i] = m->len[u];
    max_q = std::max(max_q, L);
    for (int j = 0; j < L; ++j) {
      const int t = tokens_host[(int64_t)i * max_len + j];
      WN_CHECK(t >= 0 && t < V, "decoder step: token id");
      tok.push_back(t);
      pos.push_back(j);
    }
    last[i] = qoff[i] + L - 1;
  }
  const int R = (int)tok.size();
  WN_TRY(m->stage.begin((size_t)(2 * R + 5 * n_seq + 64) * sizeof(int) + 4096));
  WN_TRY(upload_desc(m, m->r_tok, tok, s));
  WN_TRY(upload_desc(m, m->r_pos, pos, s));
  WN_TRY(upload_desc(m, m->r_qoff, qoff, s));
  WN_TRY(upload_desc(m, m->r_qlen, qlen, s));
  WN_TRY(upload_desc(m, m->r_kvoff, kvoff, s));
  WN_TRY(upload_desc(m, m->r_kvlen, kvlen, s));
  WN_TRY(upload_desc(m, m->r_tgt, last, s));
  WN_TRY(m->stage.end(s));
  WN_TRY(m->r_x.ensure((size_t)R * d * sizeof(float)));
  WN_TRY(m->r_t1.ensure((size_t)std::max(R, n_seq) * d * sizeof(float)));
  WN_TRY(m->r_t2.ensure((size_t)std::max(R, n_seq) * d * sizeof(float)));
  WN_TRY(m->r_qkv.ensure((size_t)R * 3 * d * sizeof(float)));
  WN_TRY(m->r_h.ensure((size_t)R * c.dec_ffn_dim * sizeof(float)));
  WN_TRY(m->r_logits.ensure((size_t)n_seq * V * sizeof(float)));
  WN_TRY(m->r_out.ensure((size_t)2 * n_seq * topk * sizeof(float)));
  WN_TRY(decoder_layers(m, m->left, R, n_seq, max_q, m->r_tok.as<int>(), true, s));
  // y = log_softmax(output_layer(after_norm(x[:, -1])))   decoder.py:275-281
  float* t2 = m->r_t2.as<float>();
  float* t1 = m->r_t1.as<float>();
  WN_TRY(copy_rows(m->r_x.as<float>(), d, m->r_tgt.as<int>(), t2, d, nullptr, n_seq, d, s));
  WN_TRY(ln(m->left.after, t2, t1, n_seq, d, c.norm_eps, s));
  WN_TRY(linear(m->left.out, t1, d, m->r_logits.as<float>(), V, n_seq, s));
  float* tv = m->r_out.as<float>();
  int* ti = reinterpret_cast<int*>(tv + (size_t)n_seq * topk);
  CtcRowArgs r;
  r.logits = m->r_logits.as<float>(); r.ld = V; r.M = n_seq; r.V = V; r.k = topk;
  r.blank = -1; r.blank_penalty = 0.f;
  r.topk_val = tv; r.topk_idx = ti; r.logp = nullptr; r.ld_out = V;
  WN_TRY(ctc_logsoftmax_topk(r, s));
  WN_HIP(hipMemcpyAsync(logp_host, tv, (size_t)n_seq * topk * sizeof(float),
                        hipMemcpyDeviceToHost, s));
  WN_HIP(hipMemcpyAsync(idx_host, ti, (size_t)n_seq * topk * sizeof(int),
                        hipMemcpyDeviceToHost, s));
  WN_HIP(hipStreamSynchronize(s));
  return 0;
}

// attention_beam_search (search.py:252-371) for the current batch, entirely on the
// device: one decoder row per running hypothesis and step (self-attention K/V cache
// addressed through per-hypothesis ancestor paths, cross-attention K/V projected once),
// beam bookkeeping in beam_update_kernel; the host only reads the "all ended" counter.
int wn_attention_beam_search(wn_model* m, int32_t beam, int32_t maxlen, float length_penalty,
                             int32_t* tokens_host, int32_t* lens_host, void* stream) {
  WN_CHECK(m && m->B > 0 && m->enc.p, "attention beam search: no current batch");
  WN_ENTER(m);
  PrecisionScope prec_scope(m);
  WN_CHECK(!m->left.layers.empty(), "attention beam search: the model has no attention decoder");
  WN_CHECK(beam >= 1 && beam <= 64 && maxlen >= 1 && tokens_host && lens_host,
           "attention beam search: beam_size in [1, 64], maxlen >= 1");
  hipStream_t s = (hipStream_t)stream;
  WN_HIP(hipSetDevice(m->device));
  const wn_config& c = m->cfg;
  const Decoder& D = m->left;
  const int d = c.d_model, V = c.vocab, B = m->B, N = beam, BN = B * N, Menc = m->rows;
  WN_CHECK(beam <= V, "attention beam search: beam larger than the vocabulary");
  WN_CHECK(maxlen + 1 <= c.max_pos, "attention beam search: longer than the positional table");
  const int W = maxlen + 2;                       // columns of the token / path rows
  const int nl = (int)D.layers.size();
  for (int b = 0; b < B; ++b)
    WN_CHECK(m->len[b] > 0, "attention beam search: utterance without encoder frames");
  // ---- descriptors of the cross attention: one query row per hypothesis -------------
  std::vector<int> qoff(BN), qlen(BN, 1), kvoff(BN), kvlen(BN);
  for (int r = 0; r < BN; ++r) { qoff[r] = r; kvoff[r] = m->off[r / N]; kvlen[r] = m->len[r / N]; }
  WN_TRY(m->stage.begin((size_t)(4 * BN + 64) * sizeof(int) + 4096));
  WN_TRY(upload_desc(m, m->r_qoff, qoff, s));
  WN_TRY(upload_desc(m, m->r_qlen, qlen, s));
  WN_TRY(upload_desc(m, m->r_kvoff, kvoff, s));
  WN_TRY(upload_desc(m, m->r_kvlen, kvlen, s));
  WN_TRY(m->stage.end(s));
  WN_TRY(m->r_x.ensure((size_t)BN * d * sizeof(float)));
  WN_TRY(m->r_t1.ensure((size_t)BN * d * sizeof(float)));
  WN_TRY(m->r_t2.ensure((size_t)BN * d * sizeof(float)));
  WN_TRY(m->r_qkv.ensure((size_t)BN * 3 * d * sizeof(float)));
  WN_TRY(m->r_h.ensure((size_t)BN * c.dec_ffn_dim * sizeof(float)));
  WN_TRY(m->r_logits.ensure((size_t)BN * V * sizeof(float)));
  WN_TRY(m->r_out.ensure((size_t)2 * BN * N * sizeof(float)));
  // self-attention K | V cache [layer][step][slot][2d]: sized for the steps actually run,
  // not for maxlen = T' (the reference's cache grows with the decoded length too,
  // decoder.py:226-281): starts at 32 steps and doubles, the used prefix of every layer is
  // carried over
  int cap_steps = std::min(maxlen, 32);
  size_t cache_layer = (size_t)cap_steps * BN * 2 * d;
  WN_TRY(m->ab_cache.ensure(nl * cache_layer * sizeof(float)));
  auto grow_cache = [&](int used_steps) -> int {
    const int cap2 = std::min(maxlen, cap_steps * 2);
    const size_t layer2 = (size_t)cap2 * BN * 2 * d;
    DevBuf nb;
    WN_TRY(nb.ensure(nl * layer2 * sizeof(float)));
    for (int li = 0; li < nl; ++li)
      WN_HIP(hipMemcpyAsync(nb.as<float>() + li * layer2, m->ab_cache.as<float>() + li * cache_layer,
                            (size_t)used_steps * BN * 2 * d * sizeof(float),
                            hipMemcpyDeviceToDevice, s));
    WN_HIP(hipStreamSynchronize(s));            // before the old buffer is freed
    std::swap(m->ab_cache.p, nb.p);
    std::swap(m->ab_cache.cap, nb.cap);
    cap_steps = cap2;
    cache_layer = layer2;
    return 0;
  };
  const size_t mem_layer = (size_t)Menc * 2 * d;
  WN_TRY(m->r_mem_all.ensure(nl * mem_layer * sizeof(float)));
  // state: 2 x {score, end, tok, path} + last_tok + n_done + out_tok + out_len
  const size_t n_int = (size_t)2 * (BN + BN + (size_t)BN * W * 2) + BN + 16 + (size_t)B * W + B;
  WN_TRY(m->ab_state.ensure(n_int * sizeof(int)));
  int* base = m->ab_state.as<int>();
  float* score[2]; int* endf[2]; int* tok[2]; int* path[2];
  for (int k = 0; k < 2; ++k) {
    score[k] = reinterpret_cast<float*>(base); base += BN;
    endf[k] = base; base += BN;
    tok[k] = base; base += (size_t)BN * W;
    path[k] = base; base += (size_t)BN * W;
  }
  int* last_tok = base; base += BN;
  int* n_done = base; base += 16;
  int* out_tok = base; base += (size_t)B * W;
  int* out_len = base;
  WN_TRY(attn_beam_init(BN, N, W, c.sos, score[0], endf[0], tok[0], path[0], last_tok, s));
  WN_HIP(hipMemsetAsync(n_done, 0, sizeof(int), s));
  float* x = m->r_x.as<float>();
  float* t1 = m->r_t1.as<float>();
  float* t2 = m->r_t2.as<float>();
  float* qkv = m->r_qkv.as<float>();
  float* hb = m->r_h.as<float>();
  float* tv = m->r_out.as<float>();
  int* ti = reinterpret_cast<int*>(tv + (size_t)BN * N);
  const float eps = c.norm_eps;
  int cur = 0, len = 1, done_host = 0;
  for (int i = 1; i <= maxlen; ++i) {
    if (done_host == BN) break;
    const int step = i - 1;                       // position of the newest token
    if (step >= cap_steps) WN_TRY(grow_cache(step));
    WN_TRY(attn_step_embed(last_tok, step, D.embed, D.pe, sqrtf((float)d), d, BN, x, s));
    for (int li = 0; li < nl; ++li) {
      const DecLayer& L = D.layers[li];
      WN_TRY(ln(L.n1, x, t1, BN, d, eps, s));
      WN_TRY(linear(L.self_qkv, t1, d, qkv, 3 * d, BN, s));
      WN_TRY(attn_self_step(qkv, d, c.dec_heads, BN, m->ab_cache.as<float>() + li * cache_layer,
                            step, path[cur], W, t2, s));
      WN_TRY(linear(L.self_out, t2, d, x, d, BN, s, ACT_NONE, x, d));
      WN_TRY(ln(L.n2, x, t1, BN, d, eps, s));
      WN_TRY(linear(L.src_q, t1, d, t2, d, BN, s));
      float* mem = m->r_mem_all.as<float>() + (size_t)li * mem_layer;
      if (!m->mem_cache_valid) WN_TRY(linear(L.src_kv, m->enc.as<float>(), d, mem, 2 * d, Menc, s));
      AttnArgs cx;
      cx.Q = t2; cx.ldq = d; cx.K = mem; cx.V = mem + d; cx.ldk = cx.ldv = 2 * d;
      cx.O = t1; cx.ldo = d;
      cx.q_off = m->r_qoff.as<int>(); cx.q_len = m->r_qlen.as<int>();
      cx.kv_off = m->r_kvoff.as<int>(); cx.kv_len = m->r_kvlen.as<int>();
      cx.n_seq = BN; cx.n_heads = c.dec_heads; cx.max_q_len = 1;
      cx.mask_mode = 0; cx.scale = 0.125f;
      WN_TRY(attention(cx, s));
      WN_TRY(linear(L.src_out, t1, d, x, d, BN, s, ACT_NONE, x, d));
      WN_TRY(ln(L.n3, x, t1, BN, d, eps, s));
      WN_TRY(linear(L.ff1, t1, d, hb, c.dec_ffn_dim, BN, s, ACT_RELU));
      WN_TRY(linear(L.ff2, hb, c.dec_ffn_dim, x, d, BN, s, ACT_NONE, x, d));
    }
    m->mem_cache_valid = true;
    // log_softmax(output_layer(after_norm(x))) -> the N best (log-prob, token) per row
    WN_TRY(ln(D.after, x, t1, BN, d, eps, s));
    WN_TRY(linear(D.out, t1, d, m->r_logits.as<float>(), V, BN, s));
    CtcRowArgs r;
    r.logits = m->r_logits.as<float>(); r.ld = V; r.M = BN; r.V = V; r.k = N;
    r.blank = -1; r.blank_penalty = 0.f;
    r.topk_val = tv; r.topk_idx = ti; r.logp = nullptr; r.ld_out = V;
    WN_TRY(ctc_logsoftmax_topk(r, s));
    WN_HIP(hipMemsetAsync(n_done, 0, sizeof(int), s));
    WN_TRY(attn_beam_update(B, N, i, W, c.eos, V, tv, ti, score[cur], endf[cur], tok[cur],
                            path[cur], score[cur ^ 1], endf[cur ^ 1], tok[cur ^ 1],
                            path[cur ^ 1], last_tok, n_done, s));
    cur ^= 1;
    len = i + 1;
    // "all hypotheses ended" is polled every 4th step: a step run after the end only appends
    // eos to finished hypotheses and leaves their scores alone (mask_finished_scores /
    // _preds), and the result strips eos (search.py:355-371) -- same output, 3 of 4 host
    // round trips fewer
    if ((i & 3) == 0 || i == maxlen) {
      WN_HIP(hipMemcpyAsync(&done_host, n_done, sizeof(int), hipMemcpyDeviceToHost, s));
      WN_HIP(hipStreamSynchronize(s));
    }
  }
  WN_TRY(attn_beam_finish(B, N, len, W, c.eos, length_penalty, score[cur], tok[cur], out_tok,
                          out_len, s));
  std::vector<int> ot((size_t)B * W), ol(B);
  WN_HIP(hipMemcpyAsync(ot.data(), out_tok, ot.size() * sizeof(int), hipMemcpyDeviceToHost, s));
  WN_HIP(hipMemcpyAsync(ol.data(), out_len, ol.size() * sizeof(int), hipMemcpyDeviceToHost, s));
  WN_HIP(hipStreamSynchronize(s));
  for (int b = 0; b < B; ++b) {
    lens_host[b] = std::min(ol[b], maxlen);
    for (int j = 0; j < lens_host[b]; ++j) tokens_host[(size_t)b * maxlen + j] = ot[(size_t)b * W + j];
  }
  return 0;
}

int wn_decoder_forward(wn_model* m, int32_t utt, int32_t which, int32_t n_seq,
                       const int32_t* tokens_host, const int32_t* lens_host,
                       int32_t max_len, float* logp_dev, void* stream) {
  WN_CHECK(m && m->B > 0 && m->enc.p, "decoder forward: no current batch");
  WN_ENTER(m);
  PrecisionScope prec_scope(m);
  WN_CHECK(tokens_host && lens_host && logp_dev, "decoder forward: null argument");
  WN_CHECK(utt >= 0 && utt < m->B && m->len[utt] > 0,
           "decoder forward: utterance index / no encoder frames");
  WN_CHECK(which == 0 || which == 1, "decoder forward: which must be 0 (left) or 1 (right)");
  const Decoder& D = which == 0 ? m->left : m->right;
  WN_CHECK(!D.layers.empty(), "decoder forward: the model has no such decoder");
  WN_CHECK(n_seq > 0 && max_len > 0 && max_len <= m->cfg.max_pos,
           "decoder forward: bad batch shape");
  hipStream_t s = (hipStream_t)stream;
  WN_HIP(hipSetDevice(m->device));
  const wn_config& c = m->cfg;
  const int d = c.d_model, V = c.vocab;
  const int R = n_seq * max_len;
  std::vector<int> tok(R), pos(R), qoff(n_seq), qlen(n_seq), kvoff(n_seq), kvlen(n_seq),
      slen(n_seq);
  for (int i = 0; i < n_seq; ++i) {
    WN_CHECK(lens_host[i] >= 1 && lens_host[i] <= max_len, "decoder forward: length");
    qoff[i] = i * max_len; qlen[i] = max_len; slen[i] = lens_host[i];
    kvoff[i] = m->off[utt]; kvlen[i] = m->len[utt];
    for (int j = 0; j < max_len; ++j) {
      const int t = tokens_host[(size_t)i * max_len + j];
      WN_CHECK(t >= 0 && t < V, "decoder forward: token id out of range");
      tok[(size_t)i * max_len + j] = t;
      pos[(size_t)i * max_len + j] = j;
    }
  }
  WN_TRY(m->stage.begin((size_t)(2 * R + 5 * n_seq + 64) * sizeof(int) + 4096));
  WN_TRY(upload_desc(m, m->r_tok, tok, s));
  WN_TRY(upload_desc(m, m->r_pos, pos, s));
  WN_TRY(upload_desc(m, m->r_qoff, qoff, s));
  WN_TRY(upload_desc(m, m->r_qlen, qlen, s));
  WN_TRY(upload_desc(m, m->r_kvoff, kvoff, s));
  WN_TRY(upload_desc(m, m->r_kvlen, kvlen, s));
  WN_TRY(upload_desc(m, m->r_tgt, slen, s));  // self-attention key lengths
  WN_TRY(m->stage.end(s));
  WN_TRY(m->r_x.ensure((size_t)R * d * sizeof(float)));
  WN_TRY(m->r_t1.ensure((size_t)R * d * sizeof(float)));
  WN_TRY(m->r_t2.ensure((size_t)R * d * sizeof(float)));
  WN_TRY(m->r_qkv.ensure((size_t)R * 3 * d * sizeof(float)));
  WN_TRY(m->r_h.ensure((size_t)R * c.dec_ffn_dim * sizeof(float)));
  WN_TRY(m->r_mem.ensure((size_t)m->rows * 2 * d * sizeof(float)));
  WN_TRY(m->r_logits.ensure((size_t)R * ((V + 3) / 4 * 4) * sizeof(float)));
  WN_TRY(m->r_out.ensure((size_t)2 * R * sizeof(float)));
  WN_TRY(decoder_layers(m, D, R, n_seq, max_len, m->r_tok.as<int>(), false, s,
                        m->r_tgt.as<int>()));
  float* t1 = m->r_t1.as<float>();
  WN_TRY(ln(D.after, m->r_x.as<float>(), t1, R, d, c.norm_eps, s));
  WN_TRY(linear(D.out, t1, d, m->r_logits.as<float>(), V, R, s));
  // log_softmax over the vocabulary of every row (asr_model.py:543-546)
  CtcRowArgs a;
  a.logits = m->r_logits.as<float>(); a.ld = V; a.M = R; a.V = V; a.k = 1;
  a.blank = 0; a.blank_penalty = 0.f;
  a.topk_val = m->r_out.as<float>();
  a.topk_idx = reinterpret_cast<int*>(m->r_out.as<float>() + R);
  a.logp = logp_dev; a.ld_out = V;
  return ctc_logsoftmax_topk(a, s);
}

int wn_attention_rescoring(wn_model* m, int32_t beam, const int32_t* n_hyps_host,
                           const int32_t* hyp_lens_host,
                           const int32_t* hyp_tokens_host, int32_t max_len,
                           float reverse_weight, float* l2r_logp_host,
                           float* r2l_logp_host, void* stream) {
  WN_CHECK(m && m->B > 0 && m->enc.p, "rescoring: no current batch");
  WN_ENTER(m);
  PrecisionScope prec_scope(m);
  WN_CHECK(!m->left.layers.empty(), "rescoring: the model has no attention decoder");
  WN_CHECK(n_hyps_host && hyp_lens_host && hyp_tokens_host && l2r_logp_host &&
               r2l_logp_host, "rescoring: null argument");
  hipStream_t s = (hipStream_t)stream;
  WN_HIP(hipSetDevice(m->device));
  const wn_config& c = m->cfg;
  const int B = m->B, d = c.d_model, V = c.vocab;
  const bool use_r2l = reverse_weight > 0.f && !m->right.layers.empty();
  // ---- ragged hypothesis batch: row = (utt, hyp, position) -------------------
  std::vector<int> tok, rtok, pos, tgt, rtgt, qoff, qlen, kvoff, kvlen;
  std::vector<int64_t> out_index;  // row -> index into (B, beam, max_len+1)
  int max_q = 0;
  for (int b = 0; b < B; ++b) {
    WN_CHECK(n_hyps_host[b] >= 0 && n_hyps_host[b] <= beam, "rescoring: n_hyps");
    if (n_hyps_host[b] > 0)
      WN_CHECK(m->len[b] > 0, "rescoring: utterance without encoder frames");
    for (int i = 0; i < n_hyps_host[b]; ++i) {
      const int L = hyp_lens_host[b * beam + i];
      WN_CHECK(L >= 0 && L <= max_len, "rescoring: hypothesis length");
      WN_CHECK(L + 1 <= c.max_pos, "rescoring: hypothesis longer than the positional table");
      const int32_t* h = hyp_tokens_host + ((int64_t)b * beam + i) * max_len;
      qoff.push_back((int)tok.size());
      qlen.push_back(L + 1);
      kvoff.push_back(m->off[b]);
      kvlen.push_back(m->len[b]);
      max_q = std::max(max_q, L + 1);
      for (int j = 0; j <= L; ++j) {
        // add_sos_eos (common.py:113-155): ys_in = [sos] + hyp
        const int t_in = j == 0 ? c.sos : h[j - 1];
        const int rt_in = j == 0 ? c.sos : h[L - j];  // reversed hyp (asr_model.py:491-536)
        WN_CHECK(t_in >= 0 && t_in < V && rt_in >= 0 && rt_in < V, "rescoring: token id");
        tok.push_back(t_in);
        rtok.push_back(rt_in);
        pos.push_back(j);
        tgt.push_back(j < L ? h[j] : c.eos);
        rtgt.push_back(j < L ? h[L - 1 - j] : c.eos);
        out_index.push_back(((int64_t)b * beam + i) * (max_len + 1) + j);
      }
    }
  }
  const int R = (int)tok.size(), n_seq = (int)qoff.size();
  const size_t out_n = (size_t)B * beam * (max_len + 1);
  memset(l2r_logp_host, 0, out_n * sizeof(float));
  memset(r2l_logp_host, 0, out_n * sizeof(float));
  if (R == 0) return 0;
  WN_TRY(m->stage.begin((size_t)(5 * R + 4 * n_seq + 64) * sizeof(int) + 4096));
  WN_TRY(upload_desc(m, m->r_tok, tok, s));
  WN_TRY(upload_desc(m, m->r_rtok, rtok, s));
  WN_TRY(upload_desc(m, m->r_pos, pos, s));
  WN_TRY(upload_desc(m, m->r_tgt, tgt, s));
  WN_TRY(upload_desc(m, m->r_rtgt, rtgt, s));
  WN_TRY(upload_desc(m, m->r_qoff, qoff, s));
  WN_TRY(upload_desc(m, m->r_qlen, qlen, s));
  WN_TRY(upload_desc(m, m->r_kvoff, kvoff, s));
  WN_TRY(upload_desc(m, m->r_kvlen, kvlen, s));
  WN_TRY(m->stage.end(s));
  WN_TRY(m->r_x.ensure((size_t)R * d * sizeof(float)));
  WN_TRY(m->r_t1.ensure((size_t)R * d * sizeof(float)));
  WN_TRY(m->r_t2.ensure((size_t)R * d * sizeof(float)));
  WN_TRY(m->r_qkv.ensure((size_t)R * 3 * d * sizeof(float)));
  WN_TRY(m->r_h.ensure((size_t)R * c.dec_ffn_dim * sizeof(float)));
  WN_TRY(m->r_mem.ensure((size_t)m->rows * 2 * d * sizeof(float)));
  WN_TRY(m->r_logits.ensure((size_t)R * ((V + 3) / 4 * 4) * sizeof(float)));
  WN_TRY(m->r_out.ensure((size_t)2 * R * sizeof(float)));
  float* o_l = m->r_out.as<float>();
  float* o_r = o_l + R;
  WN_TRY(run_decoder(m, m->left, R, n_seq, max_q, m->r_tok.as<int>(),
                     m->r_tgt.as<int>(), o_l, s));
  if (use_r2l)
    WN_TRY(run_decoder(m, m->right, R, n_seq, max_q, m->r_rtok.as<int>(),
                       m->r_rtgt.as<int>(), o_r, s));
  std::vector<float> hl(R), hr(R, 0.f);
  WN_HIP(hipMemcpyAsync(hl.data(), o_l, R * sizeof(float), hipMemcpyDeviceToHost, s));
  if (use_r2l)
    WN_HIP(hipMemcpyAsync(hr.data(), o_r, R * sizeof(float), hipMemcpyDeviceToHost, s));
  WN_HIP(hipStreamSynchronize(s));
  for (int r = 0; r < R; ++r) {
    l2r_logp_host[out_index[r]] = hl[r];
    r2l_logp_host[out_index[r]] = hr[r];
  }
  return 0;
}

// ---------------------------------------------------------------------------
int wn_op_gemm(const float* A, const float* W, const float* bias,
               const float* resid, float* C, int32_t M, int32_t N, int32_t K,
               float alpha, int32_t act, void* stream) {
  GemmArgs g;
  g.A = A; g.W = W; g.bias = bias; g.resid = resid; g.C = C;
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N; g.ldr = N;
  g.alpha = alpha; g.act = act;
  return gemm_f32(g, (hipStream_t)stream);
}

int wn_op_gemm_bf16(const float* A, const float* W, const float* bias,
                    const float* resid, float* C, int32_t M, int32_t N, int32_t K,
                    float alpha, int32_t act, void* stream) {
  const int saved = t_gemm_prec;
  t_gemm_prec = PREC_BF16;
  const int r = wn_op_gemm(A, W, bias, resid, C, M, N, K, alpha, act, stream);
  t_gemm_prec = saved;
  return r;
}

int wn_op_gemm_bf16_stored(const float* A, const float* W, const float* bias,
                           const float* resid, void* C, int32_t M, int32_t N, int32_t K,
                           float alpha, int32_t act, int32_t c_bf16, void* stream) {
  // test hook of the bf16-storage GEMM: A and W are converted to bf16 images in
  // scratch buffers first (the model path gets them from its producers / the
  // converted weight slab)
  WN_CHECK(A && W && C && M > 0 && N > 0 && K > 0, "gemm(bf16 stored): null / empty");
  WN_CHECK(K % 32 == 0, "gemm: K must be a multiple of 32");
  static thread_local DevBuf a16, w16;
  hipStream_t s = (hipStream_t)stream;
  WN_TRY(a16.ensure((size_t)M * K * 2));
  WN_TRY(w16.ensure((size_t)N * K * 2));
  WN_TRY(convert_f32_to_bf16(A, a16.p, (int64_t)M * K, s));
  WN_TRY(convert_f32_to_bf16(W, w16.p, (int64_t)N * K, s));
  GemmArgs g;
  g.A = a16.as<float>(); g.W = W; g.bias = bias; g.resid = resid;
  g.C = reinterpret_cast<float*>(C);
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N; g.ldr = N;
  g.alpha = alpha; g.act = act; g.a_bf16 = true; g.c_bf16 = c_bf16 != 0;
  return gemm_bf16_stored(g, w16.p, s);
}

int wn_op_gemm_lowp(const void* A, const void* W, const void* a_scale, const void* w_scale,
                    const float* bias, const float* resid, void* C, void* c_scale,
                    int32_t M, int32_t N, int32_t K, float alpha, int32_t act,
                    int32_t c_mode, int32_t dtype, void* stream) {
  WN_CHECK(A && W && C && M > 0 && N > 0 && K > 0, "gemm(lowp): null / empty");
  WN_CHECK(K % 32 == 0, "gemm: K must be a multiple of 32");
  GemmArgs g;
  g.A = reinterpret_cast<const float*>(A); g.W = nullptr; g.bias = bias; g.resid = resid;
  g.C = reinterpret_cast<float*>(C);
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N; g.ldr = N;
  g.alpha = alpha; g.act = act;
  if (dtype == 1) {
    WN_CHECK(c_mode == 0 || c_mode == 1, "gemm(lowp): bf16 operands give fp32 / bf16 C");
    g.a_bf16 = true; g.c_bf16 = c_mode == 1;
    return gemm_bf16_stored(g, W, (hipStream_t)stream);
  }
  if (dtype == 2) {
    WN_CHECK(c_mode == 0 || c_mode == 2, "gemm(lowp): MXFP8 operands give fp32 / MXFP8 C");
    g.fp8 = true; g.c_mx = c_mode == 2;
    g.a_scale = reinterpret_cast<const unsigned*>(a_scale); g.a_scale_pitch = M;
    g.w_scale = reinterpret_cast<const unsigned*>(w_scale); g.w_scale_pitch = N;
    g.c_scale = reinterpret_cast<unsigned*>(c_scale); g.c_scale_pitch = M;
    return gemm_mxfp8(g, W, (hipStream_t)stream);
  }
  set_error("gemm(lowp): unknown dtype");
  return -1;
}

int wn_op_mx_quantize(const float* x, int32_t rows, int32_t K, void* q, void* scale,
                      void* stream) {
  WN_CHECK(x && q && scale && rows > 0 && K > 0 && K % 128 == 0,
           "mx_quantize: null / empty / K % 128");
  return mx_quantize(x, K, rows, K, q, reinterpret_cast<unsigned*>(scale), rows,
                     (hipStream_t)stream);
}

int wn_op_ffn_fused(const float* X, const float* W1, const float* b1, const float* W2,
                    const float* b2, float* x, const float* ln_w, const float* ln_b,
                    float* y, int32_t M, int32_t D, int32_t F, int32_t act, float alpha,
                    float eps, void* stream) {
  WN_CHECK(X && W1 && b1 && W2 && b2 && x && ln_w && ln_b && y, "ffn_fused: null argument");
  WN_CHECK(M > 0 && (D == 256 || D == 512) && F > 0 && F % 64 == 0, "ffn_fused: shape");
  const int S = ffn_fused_split(M, D, F);
  WN_CHECK(S > 0, "ffn_fused: hidden size cannot be split for this M");
  static thread_local DevBuf part;
  WN_TRY(part.ensure((size_t)S * M * D * sizeof(float)));
  FfnArgs a;
  a.X = X; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.P = part.as<float>();
  a.M = M; a.D = D; a.F = F; a.S = S; a.act = act;
  WN_TRY(ffn_fused(a, (hipStream_t)stream));
  return ffn_reduce_ln(x, part.as<float>(), S, b2, alpha, ln_w, ln_b, nullptr, nullptr, y, M,
                       D, eps, 0, (hipStream_t)stream);
}

int wn_op_gemm_x6(const float* A, const float* W, const float* bias, const float* resid,
                  float* C, int32_t M, int32_t N, int32_t K, float alpha, int32_t act,
                  int32_t bm, int32_t reps, void* stream) {
  WN_CHECK(A && W && C && M > 0 && N > 0 && K > 0 && K % 16 == 0 && N % 4 == 0,
           "gemm_x6: shape");
  hipStream_t s = (hipStream_t)stream;
  static thread_local DevBuf a3, w3;
  WN_TRY(w3.ensure(x6_bytes(N, K)));
  WN_TRY(x6_split(W, N, K, K, w3.as<char>(), s));
  X6Args a;
  if (g_x6_af32 != 0 && (int64_t)M * K * 4 < ((int64_t)1 << 31)) {
    a.A = A; a.lda = K; a.a_bytes = (int64_t)M * K * 4;      // split in registers
  } else {
    WN_TRY(a3.ensure(x6_bytes(M, K)));
    WN_TRY(x6_split(A, M, K, K, a3.as<char>(), s));
    a.A3 = a3.as<char>();
  }
  a.B3 = w3.as<char>(); a.M = M; a.N = N; a.K = K; a.bm = bm;
  if (bm == 120) { a.bm = 128; a.nw = 8; }   // micro-benchmark: the 8-wave 128-row tile
  if (bm >= 129 && bm <= 132) {      // 4-wave 128-row tiles with priorities (130+)
    a.bm = 128; a.nw = 4;
    if (bm == 130) a.prio_split = cdiv(M, 128) * cdiv(N, 256) / 2;
    if (bm == 131) a.prio_split = -1;
    if (bm == 132) a.prio_split = -2;
  }
  a.bias = bias; a.resid = resid; a.ldr = N; a.alpha = alpha; a.act = act; a.C = C; a.ldc = N;
  for (int r = 0; r < (reps > 0 ? reps : 1); ++r) WN_TRY(gemm_x6(a, s));
  return 0;
}

int wn_op_ffn_x6(const float* X, const float* W1, const float* b1, const float* W2,
                 const float* b2, float* x, const float* ln_w, const float* ln_b, float* y,
                 int32_t M, int32_t D, int32_t F, int32_t act, float alpha, float eps,
                 int32_t reps, void* stream) {
  WN_CHECK(X && W1 && b1 && W2 && b2 && x && ln_w && ln_b && y, "ffn_x6: null argument");
  WN_CHECK(M > 0 && (D == 256 || D == 512) && F > 0 && F % 64 == 0, "ffn_x6: shape");
  hipStream_t s = (hipStream_t)stream;
  static thread_local DevBuf x3, w13, w23, h3, part;
  if (g_ffn_x6f != 0 && g_x6_af32 == 0 && ffn_x6f_supported(M, D, F, act)) {
    // hidden tensor on chip (ffn_x6f.hip)
    FfnX6Args a;
    a.S = ffn_x6f_split(M, F);
    WN_TRY(w13.ensure(x6_bytes(F, D)));
    WN_TRY(w23.ensure(x6_bytes(D, F)));
    WN_TRY(part.ensure((size_t)a.S * M * D * sizeof(float)));
    WN_TRY(x6_split(W1, F, D, D, w13.as<char>(), s));
    WN_TRY(x6_split_perm(W2, D, F, F, w23.as<char>(), s));
    a.X = X; a.ldx = D; a.W13 = w13.as<char>(); a.W2p = w23.as<char>(); a.b1 = b1;
    a.P = part.as<float>(); a.M = M; a.D = D; a.F = F; a.act = act;
    for (int r = 0; r < (reps > 0 ? reps : 1); ++r) WN_TRY(ffn_x6f(a, s));
    return ffn_reduce_ln(x, part.as<float>(), a.S, b2, alpha, ln_w, ln_b, nullptr, nullptr, y, M,
                         D, eps, 0, s);
  }
  const int S = ffn_x6_split(M, F);
  WN_TRY(x3.ensure(x6_bytes(M, D)));
  WN_TRY(w13.ensure(x6_bytes(F, D)));
  WN_TRY(w23.ensure(x6_bytes(D, F)));
  WN_TRY(h3.ensure(x6_bytes(M, F)));
  WN_TRY(part.ensure((size_t)S * M * D * sizeof(float)));
  WN_TRY(x6_split(W1, F, D, D, w13.as<char>(), s));
  WN_TRY(x6_split(W2, D, F, F, w23.as<char>(), s));
  const bool af32 = g_x6_af32 != 0 && (int64_t)M * F * 4 < ((int64_t)1 << 31);
  static thread_local DevBuf hf;
  if (af32) WN_TRY(hf.ensure((size_t)M * F * sizeof(float)));
  for (int r = 0; r < (reps > 0 ? reps : 1); ++r) {
    X6Args g1, g2;
    g1.B3 = w13.as<char>(); g1.M = M; g1.N = F; g1.K = D; g1.bias = b1; g1.act = act;
    g2.B3 = w23.as<char>(); g2.M = M; g2.N = D; g2.K = F;
    g2.epi = 1; g2.ksplit = S; g2.C = part.as<float>();
    if (af32) {
      g1.A = X; g1.lda = D; g1.a_bytes = (int64_t)M * D * 4;
      g1.epi = 0; g1.C = hf.as<float>(); g1.ldc = F;
      g2.A = hf.as<float>(); g2.lda = F; g2.a_bytes = (int64_t)M * F * 4;
    } else {
      WN_TRY(x6_split(X, M, D, D, x3.as<char>(), s));
      g1.A3 = x3.as<char>(); g1.epi = 2; g1.C3 = h3.as<char>();
      g2.A3 = h3.as<char>();
    }
    WN_TRY(gemm_x6(g1, s));
    WN_TRY(gemm_x6(g2, s));
    if (r + 1 < reps) continue;      // timing loops: the residual update only once
    WN_TRY(ffn_reduce_ln(x, part.as<float>(), S, b2, alpha, ln_w, ln_b, nullptr, nullptr, y,
                         M, D, eps, 0, s));
  }
  return 0;
}

int wn_op_gemm_x6r(const float* A, const float* W, const float* bias, float* x_inout,
                   const float* ln_w, const float* ln_b, float* y, float* C, int32_t M,
                   int32_t N, int32_t epi, float alpha, float eps, int32_t reps, void* stream) {
  WN_CHECK(A && W && M > 0 && gemm_x6r_supported(M, N, 256, epi), "gemm_x6r: shape");
  hipStream_t s = (hipStream_t)stream;
  static thread_local DevBuf w3;
  WN_TRY(w3.ensure(x6_bytes(N, 256)));
  WN_TRY(x6_split(W, N, 256, 256, w3.as<char>(), s));
  X6RArgs a;
  a.A = A; a.lda = 256; a.W3 = w3.as<char>(); a.bias = bias; a.M = M; a.N = N; a.epi = epi;
  a.C = C; a.ldc = N; a.resid = x_inout; a.ldr = N; a.alpha = alpha; a.x_out = x_inout;
  a.ldx = N; a.ln_w = ln_w; a.ln_b = ln_b; a.eps = eps; a.y = y; a.ldy = N;
  for (int r = 0; r < (reps > 0 ? reps : 1); ++r) WN_TRY(gemm_x6r(a, s));
  return 0;
}

int wn_op_log_add(const double* a_dev, const double* b_dev, double* out_dev,
                  int32_t n, void* stream) {
  return log_add_pairs(a_dev, b_dev, out_dev, n, (hipStream_t)stream);
}

int wn_op_layernorm(const float* x, const float* w, const float* b, float* y,
                    int32_t M, int32_t D, float eps, void* stream) {
  return layernorm(x, D, w, b, y, D, M, D, eps, (hipStream_t)stream);
}

namespace {
int gcd_int(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }
}  // namespace

int64_t wn_resample_length(int64_t n_in, int32_t orig_freq, int32_t new_freq) {
  if (n_in <= 0 || orig_freq <= 0 || new_freq <= 0) return 0;
  const int g = gcd_int(orig_freq, new_freq);
  const int64_t o = orig_freq / g, n = new_freq / g;
  return (n * n_in + o - 1) / o;  // ceil(new * length / orig)
}

int wn_resample(wn_model* m, const float* pcm_dev, int64_t n_in, int32_t orig_freq,
                int32_t new_freq, float* out_dev, int64_t n_out, void* stream) {
  WN_CHECK(m && pcm_dev && out_dev, "wn_resample: null argument");
  WN_ENTER(m);
  WN_CHECK(orig_freq > 0 && new_freq > 0 && n_in > 0, "wn_resample: bad rate or length");
  WN_CHECK(n_out == wn_resample_length(n_in, orig_freq, new_freq),
           "wn_resample: n_out must be wn_resample_length(n_in, orig, new)");
  hipStream_t s = (hipStream_t)stream;
  WN_HIP(hipSetDevice(m->device));
  if (orig_freq == new_freq) {  // Resample.forward returns the input unchanged
    WN_HIP(hipMemcpyAsync(out_dev, pcm_dev, (size_t)n_in * sizeof(float),
                          hipMemcpyDeviceToDevice, s));
    return 0;
  }
  const int g = gcd_int(orig_freq, new_freq);
  const int orig = orig_freq / g, nnew = new_freq / g;
  // sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99 (the defaults of
  // torchaudio.transforms.Resample); taps in fp64, stored fp32
  const double lpw = 6.0, rolloff = 0.99;
  const double base = std::min(orig, nnew) * rolloff;
  const int width = (int)std::ceil(lpw * orig / base);
  const int K = 2 * width + orig;
  std::shared_ptr<DevBuf>& buf = (*m->rs_taps)[{orig, nnew}];
  if (!buf) {
    std::vector<float> taps((size_t)nnew * K);
    const double pi = 3.14159265358979323846;
    for (int i = 0; i < nnew; ++i) {
      for (int k = 0; k < K; ++k) {
        double t = (-(double)i / nnew + (double)(k - width) / orig) * base;
        t = std::min(std::max(t, -lpw), lpw);
        const double c = std::cos(t * pi / lpw / 2.0);
        const double win = c * c;
        const double tp = t * pi;
        const double sinc = tp == 0.0 ? 1.0 : std::sin(tp) / tp;
        taps[(size_t)i * K + k] = (float)(sinc * win * (base / orig));
      }
    }
    auto nb = std::make_shared<DevBuf>();
    WN_TRY(nb->ensure(taps.size() * sizeof(float)));
    WN_HIP(hipMemcpy(nb->p, taps.data(), taps.size() * sizeof(float),
                     hipMemcpyHostToDevice));
    buf = nb;
  }
  return resample_sinc(pcm_dev, n_in, buf->as<float>(), K, width, orig, nnew, out_dev,
                       n_out, s);
}

int wn_fbank(wn_model* m, const float* pcm_dev, const int64_t* sample_off_host,
             int32_t B, float* feats_dev, int32_t max_frames,
             int32_t* n_frames_host, void* stream) {
  WN_CHECK(m && pcm_dev && sample_off_host && feats_dev && n_frames_host && B > 0,
           "wn_fbank: bad argument");
  WN_ENTER(m);
  WN_CHECK(m->fbank_ok, "wn_fbank: no Kaldi fbank for this feature dimension "
                        "(Whisper models use log-mel, processor.py:320-369)");
  hipStream_t s = (hipStream_t)stream;
  WN_HIP(hipSetDevice(m->device));
  std::vector<int> nfr(B);
  std::vector<int64_t> off(B);
  for (int b = 0; b < B; ++b) {
    const int64_t n = sample_off_host[b + 1] - sample_off_host[b];
    WN_CHECK(n >= 0, "wn_fbank: sample offsets must be non-decreasing");
    nfr[b] = n < 400 ? 0 : (int)(1 + (n - 400) / 160);   // fbank.h:254-255
    WN_CHECK(nfr[b] <= max_frames, "wn_fbank: max_frames too small");
    off[b] = sample_off_host[b];
    n_frames_host[b] = nfr[b];
  }
  if (max_frames == 0) return 0;
  WN_TRY(m->stage.begin((size_t)B * 16 + 1024));
  WN_TRY(m->stage.put(m->fb_off, off.data(), off.size() * sizeof(int64_t), s));
  WN_TRY(m->stage.put(m->fb_nfr, nfr.data(), nfr.size() * sizeof(int), s));
  WN_TRY(m->stage.end(s));
  FbankArgs a;
  a.pcm = pcm_dev; a.sample_off = m->fb_off.as<int64_t>();
  a.n_frames = m->fb_nfr.as<int>(); a.B = B; a.max_frames = max_frames;
  a.n_mel = m->cfg.feat_dim; a.window = m->fb_window; a.twiddle = m->fb_twiddle;
  const int* tab = m->fb_tab_i->as<int>();
  a.mel_start = tab; a.mel_len = tab + a.n_mel; a.mel_off = tab + 2 * a.n_mel;
  a.mel_w = m->fb_mel_w; a.feats = feats_dev;
  return fbank_kaldi(a, s);
}

namespace {
// librosa.filters.mel(sr=16000, n_fft=400, n_mels) (slaney scale + norm), the
// matrix processor.py:360-361 multiplies with (librosa is third party: its
// published algorithm is restated; the test oracle restates it independently in
// numpy).  Row-major [n_mels][LOGMEL_K2].
std::vector<float> slaney_mel_matrix(int n_mels) {
  const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, logstep = log(6.4) / 27.0;
  const double min_log_mel = min_log_hz / f_sp;
  auto hz2mel = [&](double f) {
    return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp;
  };
  auto mel2hz = [&](double mm) {
    return mm >= min_log_mel ? min_log_hz * exp(logstep * (mm - min_log_mel)) : f_sp * mm;
  };
  const int nb = 201;
  std::vector<double> mel_f(n_mels + 2);
  const double m_lo = hz2mel(0.0), m_hi = hz2mel(8000.0);
  for (int i = 0; i < n_mels + 2; ++i)
    mel_f[i] = mel2hz(m_lo + (m_hi - m_lo) * i / (double)(n_mels + 1));
  std::vector<float> w((size_t)n_mels * LOGMEL_K2, 0.f);
  for (int i = 0; i < n_mels; ++i) {
    const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
    for (int k = 0; k < nb; ++k) {
      const double f = 8000.0 * k / 200.0;
      const double lower = (f - mel_f[i]) / (mel_f[i + 1] - mel_f[i]);
      const double upper = (mel_f[i + 2] - f) / (mel_f[i + 2] - mel_f[i + 1]);
      const double v = std::max(0.0, std::min(lower, upper));
      w[(size_t)i * LOGMEL_K2 + k] = (float)(v * enorm);
    }
  }
  return w;
}
}  // namespace

int wn_log_mel(wn_model* m, const float* pcm_dev, const int64_t* sample_off_host,
               int32_t B, int32_t n_mels, float* feats_dev, int32_t max_frames,
               int32_t* n_frames_host, void* stream) {
  WN_CHECK(m && pcm_dev && sample_off_host && feats_dev && n_frames_host && B > 0,
           "wn_log_mel: bad argument");
  WN_ENTER(m);
  WN_CHECK(n_mels >= 1 && n_mels <= 256, "wn_log_mel: num_mel_bins");
  hipStream_t s = (hipStream_t)stream;
  WN_HIP(hipSetDevice(m->device));
  std::vector<int> nfr(B), foff(B), row_utt;
  std::vector<int64_t> off(B + 1);
  int rows = 0;
  for (int b = 0; b < B; ++b) {
    const int64_t n = sample_off_host[b + 1] - sample_off_host[b];
    // torch.stft(center=True) reflects n_fft/2 samples: needs n > 200
    WN_CHECK(n > 200, "wn_log_mel: an utterance needs more than 200 samples");
    nfr[b] = (int)(n / 160);          // 1 + n // hop frames, the last one dropped
    WN_CHECK(nfr[b] <= max_frames, "wn_log_mel: max_frames too small");
    off[b] = sample_off_host[b];
    foff[b] = rows;
    rows += nfr[b];
    n_frames_host[b] = nfr[b];
    for (int t = 0; t < nfr[b]; ++t) row_utt.push_back(b);
  }
  off[B] = sample_off_host[B];
  if (max_frames == 0) return 0;
  // ---- tables (once) ---------------------------------------------------------
  if (!m->lm_dft->p) {
    // [402][416] cos / -sin rows, then the periodic hann window [400]
    std::vector<float> t((size_t)LOGMEL_NS * LOGMEL_K1 + 400, 0.f);
    for (int k = 0; k <= 200; ++k)
      for (int n = 0; n < 400; ++n) {
        const double ph = 2.0 * M_PI * (double)((k * n) % 400) / 400.0;
        t[(size_t)k * LOGMEL_K1 + n] = (float)cos(ph);
        t[(size_t)(201 + k) * LOGMEL_K1 + n] = (float)-sin(ph);
      }
    for (int n = 0; n < 400; ++n)
      t[(size_t)LOGMEL_NS * LOGMEL_K1 + n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * n / 400.0));
    WN_TRY(m->lm_dft->ensure(t.size() * sizeof(float)));
    WN_HIP(hipMemcpy(m->lm_dft->p, t.data(), t.size() * sizeof(float),
                     hipMemcpyHostToDevice));
  }
  std::shared_ptr<DevBuf>& melw = (*m->lm_mel)[n_mels];
  if (!melw) {
    melw = std::make_shared<DevBuf>();
    const std::vector<float> w = slaney_mel_matrix(n_mels);
    WN_TRY(melw->ensure(w.size() * sizeof(float)));
    WN_HIP(hipMemcpy(melw->p, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  if (rows == 0) {
    WN_HIP(hipMemsetAsync(feats_dev, 0, (size_t)B * max_frames * n_mels * sizeof(float), s));
    return 0;
  }
  WN_TRY(m->stage.begin((size_t)B * 32 + (size_t)rows * 4 + 4096));
  WN_TRY(m->stage.put(m->lm_off, off.data(), off.size() * sizeof(int64_t), s));
  WN_TRY(m->stage.put(m->lm_foff, foff.data(), foff.size() * sizeof(int), s));
  WN_TRY(m->stage.put(m->lm_nfr, nfr.data(), nfr.size() * sizeof(int), s));
  WN_TRY(m->stage.put(m->lm_rowutt, row_utt.data(), row_utt.size() * sizeof(int), s));
  WN_TRY(m->stage.end(s));
  WN_TRY(m->lm_frames.ensure((size_t)rows * LOGMEL_K1 * sizeof(float)));
  WN_TRY(m->lm_spec.ensure((size_t)rows * LOGMEL_NS * sizeof(float)));
  WN_TRY(m->lm_pw.ensure((size_t)rows * LOGMEL_K2 * sizeof(float)));
  WN_TRY(m->lm_melout.ensure((size_t)rows * n_mels * sizeof(float)));
  WN_TRY(m->lm_umax.ensure((size_t)B * sizeof(float)));
  LogMelArgs a;
  a.pcm = pcm_dev; a.sample_off = m->lm_off.as<int64_t>();
  a.row_utt = m->lm_rowutt.as<int>(); a.frame_off = m->lm_foff.as<int>();
  a.window = m->lm_dft->as<float>() + (size_t)LOGMEL_NS * LOGMEL_K1;
  a.frames = m->lm_frames.as<float>();
  WN_TRY(logmel_frames(a, rows, s));
  GemmArgs g1;  // DFT: [rows, 416] x [402, 416]^T
  g1.A = m->lm_frames.as<float>(); g1.W = m->lm_dft->as<float>();
  g1.C = m->lm_spec.as<float>(); g1.M = rows; g1.N = LOGMEL_NS; g1.K = LOGMEL_K1;
  g1.lda = LOGMEL_K1; g1.ldc = LOGMEL_NS;
  WN_TRY(gemm_f32(g1, s));
  WN_TRY(logmel_power(m->lm_spec.as<float>(), m->lm_pw.as<float>(), rows, s));
  GemmArgs g2;  // mel: [rows, 224] x [n_mels, 224]^T
  g2.A = m->lm_pw.as<float>(); g2.W = melw->as<float>();
  g2.C = m->lm_melout.as<float>(); g2.M = rows; g2.N = n_mels; g2.K = LOGMEL_K2;
  g2.lda = LOGMEL_K2; g2.ldc = n_mels;
  WN_TRY(gemm_f32(g2, s));
  return logmel_finish(m->lm_melout.as<float>(), n_mels, m->lm_foff.as<int>(),
                       m->lm_nfr.as<int>(), m->lm_umax.as<float>(), B, max_frames,
                       feats_dev, s);
}

}  // extern "C"
