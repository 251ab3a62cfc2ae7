// Engine of libwenet_amd: the launch sequences of the feature -> encoder -> CTC-head path
// (subsampling, Conformer / Transformer layers, streaming chunks), the GEMM routing
// (v_mfma_f32 / six-product / bf16 / MXFP8 kernels) and the weight plane images.  The state
// it works on is model_state.h; the C ABI over it is cabi.hip.
#include <algorithm>
#include "model_state.h"

namespace wn {

static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }
const char* last_error_cstr() { return g_error.c_str(); }

}  // namespace wn

thread_local const std::map<const float*, wn_model::MxW>* t_mx = nullptr;
// plane images of the current model's weights and its activation-image scratch: linear()
// routes the large fp32 GEMMs to the six-product kernel through them (gemm_x6.hip)
thread_local const std::map<const float*, const void*>* t_x6 = nullptr;
thread_local DevBuf* t_x6_a = nullptr;

// ---- tuning knobs (tune.h) --------------------------------------------------------------------
namespace wn {
Tune g_tune_default;
thread_local const Tune* t_tune = nullptr;

int* tune_field(Tune& t, const std::string& key) {
#define X(name, dflt) if (key == #name) return &t.name;
  WN_TUNE_KEYS(X)
#undef X
  return nullptr;
}

int tune_check(const std::string& key, int32_t value, const char* who) {
  if (value == TUNE_INHERIT) return 0;
#ifndef WN_ABLATION
  if (key == "x6_probe" && (value & ~12) != 0) {
    set_error(std::string(who) + ": x6_probe 1 / 2 (no MFMAs / no DMA) need a WN_ABLATION build");
    return -1;
  }
  if (key == "ffn_x6f_var" && value != 0 && value != 25088) {
    set_error(std::string(who) + ": ffn_x6f_var variants other than the clock-stamp form "
              "(25088) need a WN_ABLATION build");
    return -1;
  }
  if (key == "attn_x6_var" && value != 0) {
    set_error(std::string(who) + ": attn_x6_var needs a WN_ABLATION build");
    return -1;
  }
  if (key == "ffn_x6f_ring" && value != 3) {
    set_error(std::string(who) + ": ffn_x6f_ring needs a WN_ABLATION build");
    return -1;
  }
#endif
  return 0;
}

void tune_resolve(const Tune& ovr, Tune* eff) {
#define X(name, dflt) eff->name = ovr.name != TUNE_INHERIT ? ovr.name : g_tune_default.name;
  WN_TUNE_KEYS(X)
#undef X
}

Tune tune_all_inherit() {
  Tune t;
#define X(name, dflt) t.name = TUNE_INHERIT;
  WN_TUNE_KEYS(X)
#undef X
  return t;
}
}  // namespace wn


// bf16-storage form of the bf16 mode: LayerNorm output, FFN hidden and attention
// context are written as bf16 (their only consumers are GEMMs that round them to
// bf16 first thing), the GEMMs read the bf16 image of the weight slab.
bool bf16_store_active() {
  return t_gemm_prec == PREC_BF16 && tune().bf16_store != 0 && tune().attn_bf16 != 0 &&
         t_wslab_bf16 != nullptr;
}

int upload_desc(wn_model* m, DevBuf& buf, const std::vector<int>& v,
                hipStream_t s) {
  return m->stage.put(buf, v.data(), v.size() * sizeof(int), s);
}

// a_bf16 / c_bf16: A / C are bf16 matrices in the same buffers (lda / ldc stay the
// element counts) -- only under bf16_store_active().
// linear(): GEMMs from this many 0.1 GFLOP on go to the six-product kernel through a split pass
static thread_local bool t_linear_took_x6 = false;   // what the last linear() of this thread ran on
constexpr int g_x6_linear_min = 60;   // (40 / 20 measured slower at configs 3 / 4, r05l)

int linear(const Linear& l, const float* A, int lda, float* C, int ldc, int M,
           hipStream_t s, int act, const float* resid, int ldr, float alpha, bool glu, bool a_bf16,
           bool c_bf16) {
  // Large fp32 GEMMs (the d = 512 encoders' projections, the decoders' GEMMs over B x N x L
  // rows): split A into planes (one pass, 4 B in / 6 B out) and run the six-product kernel
  // -- worth it from ~6 GFLOP on, where the split is a few per cent of the GEMM it halves.
  if (t_gemm_prec == PREC_F32 && tune().gemm_x6 != 0 && tune().x6_linear != 0 && t_x6 && t_x6_a && !glu &&
      !a_bf16 && !c_bf16 && l.in % 16 == 0 && l.out % 4 == 0 && lda % 4 == 0 && ldc % 4 == 0 &&
      (resid == nullptr || ldr % 4 == 0) && M >= 512 &&
      2.0 * M * (double)l.out * l.in >= 1e8 * g_x6_linear_min) {
    auto it = t_x6->find(l.w);
    if (it != t_x6->end()) {
      t_linear_took_x6 = true;
      WN_TRY(t_x6_a->ensure(x6_bytes(M, l.in)));
      WN_TRY(x6_split(A, M, l.in, lda, t_x6_a->as<char>(), s));
      X6Args x;
      x.A3 = t_x6_a->as<char>(); x.B3 = it->second; x.M = M; x.N = l.out; x.K = l.in;
      x.epi = 0; x.bias = l.b; x.resid = resid; x.ldr = ldr; x.alpha = alpha; x.act = act;
      x.C = C; x.ldc = ldc;
      return gemm_x6(x, s);
    }
  }
  t_linear_took_x6 = false;
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
  if (!m->prof_on || (m->prof_seq++ % m->prof_stride) != 0)
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
  // (the bracket also holds the plane-split pass of A when linear() took the six-product route)
  m->prof_kernel = t_linear_took_x6
                       ? "x6_split + gemm_x6_kernel (FFN w_1 + act through linear(), six bf16 plane products)"
                       : "gemm (FFN w_1)";
  return 0;
}

int ln(const Norm& n, const float* x, float* y, int M, int D, float eps,
       hipStream_t s, bool y_bf16) {
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
        t256 >= tune().fp8_min_tiles && gemm_bf16p_supported(p1) && gemm_bf16p_supported(p2)) {
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
  const bool bracket = m->prof_on && (m->prof_seq++ % m->prof_stride) == 0;
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
                                    ws.push_back(&L.pw2); ws.push_back(&L.pw1); }
  for (const Decoder* D : {&m->left, &m->right})
    for (const auto& L : D->layers) {
      ws.push_back(&L.self_qkv); ws.push_back(&L.self_out); ws.push_back(&L.src_q);
      ws.push_back(&L.src_kv); ws.push_back(&L.src_out); ws.push_back(&L.ff1);
      ws.push_back(&L.ff2);
    }
  // the Transformer encoder of the Whisper configuration (round 3): its fp32 mode goes through
  // linear() -> split pass + six-product GEMM like every other large fp32 GEMM
  for (const auto& L : m->tf_layers) { ws.push_back(&L.qkv); ws.push_back(&L.out);
                                       ws.push_back(&L.ff1); ws.push_back(&L.ff2); }
  if (m->conv2.w) ws.push_back(&m->conv2);   // [d][(ky*3+kx)*d + c]: 16-channel k blocks per tap
  if (m->sub_out.w) ws.push_back(&m->sub_out);   // K slices straight from conv2's fp32 output
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
  // QKV projections of 4-head / d_model-256 encoders, rows regrouped per head (gemm_x6r.hip
  // epi 4: wave h of a row block owns [Q_h | K_h | V_h]): new row h 192 + part 64 + j = old row
  // part 256 + h 64 + j.  One fp32 staging copy, then the ordinary split
  auto qbuf = std::make_shared<DevBuf>();
  auto qat = std::make_shared<std::map<const float*, std::pair<const void*, const float*>>>();
  if (m->cfg.d_model == 256 && m->cfg.n_heads == 4) {
    size_t n_q = 0;
    for (const auto& L : m->layers)
      if (L.qkv.w && L.qkv.b && L.qkv.out == 768 && L.qkv.in == 256 && L.pos_tab) ++n_q;
    if (n_q > 0) {
      const size_t img = x6_bytes(768, 256), per = img + 768 * sizeof(float);
      DevBuf stage;
      WN_TRY(stage.ensure((size_t)768 * 256 * sizeof(float)));
      WN_TRY(qbuf->ensure(n_q * per));
      char* q = qbuf->as<char>();
      for (const auto& L : m->layers) {
        if (!(L.qkv.w && L.qkv.b && L.qkv.out == 768 && L.qkv.in == 256 && L.pos_tab) ||
            qat->count(L.qkv.w))
          continue;
        float* qb = reinterpret_cast<float*>(q + img);
        for (int h = 0; h < 4; ++h)
          for (int part = 0; part < 3; ++part) {
            const size_t nr = (size_t)h * 192 + part * 64, orow = (size_t)part * 256 + h * 64;
            WN_HIP(hipMemcpyAsync(stage.as<float>() + nr * 256, L.qkv.w + orow * 256,
                                  64 * 256 * sizeof(float), hipMemcpyDeviceToDevice, nullptr));
            WN_HIP(hipMemcpyAsync(qb + nr, L.qkv.b + orow, 64 * sizeof(float),
                                  hipMemcpyDeviceToDevice, nullptr));
          }
        WN_TRY(x6_split(stage.as<float>(), 768, 256, 256, q, nullptr));
        (*qat)[L.qkv.w] = {q, qb};
        q += per;
      }
      WN_HIP(hipStreamSynchronize(nullptr));   // `stage` goes away with this scope
    }
  }
  m->weights_x6q = qbuf;
  m->x6q_at = qat;
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

// A vocabulary-sized layer (N = V, any V) into a logits buffer whose rows have the pitch ldc
// (>= V rounded up to 4): the six-product GEMM when the layer has a plane image (and, for V % 4 !=
// 0, a padded bias), else linear().
int vocab_linear(wn_model* m, const Linear& l, const float* A, int lda, float* C, int ldc,
                 int M, hipStream_t s) {
  const int V = l.out, V4 = (V + 3) / 4 * 4;
  WN_CHECK(ldc >= V4 && ldc % 4 == 0, "vocab_linear: pitch");
  const void* w6 = nullptr;
  const float* bias = l.b;
  if (t_gemm_prec == PREC_F32 && tune().gemm_x6 != 0 && tune().x6_linear != 0 && m->x6_at &&
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
  if (!w6) return linear(l, A, lda, C, ldc, M, s);
  WN_TRY(m->x6_lin.ensure(x6_bytes(M, l.in)));
  WN_TRY(x6_split(A, M, l.in, lda, m->x6_lin.as<char>(), s));
  X6Args x;
  x.A3 = m->x6_lin.as<char>(); x.B3 = w6; x.M = M; x.N = V4; x.K = l.in;
  x.epi = 0; x.bias = bias; x.C = C; x.ldc = ldc;
  return gemm_x6(x, s);
}

// hidden split of the x6 FFN's second GEMM: K slices so that 128-row tiles x slices fill
// the CUs once
int ffn_x6_split(int M, int F) {
  int S = 1;
  while (S < 16 && cdiv(M, 128) * (S * 2) <= 256 && (F / 16) % (S * 2) == 0) S *= 2;
  return S;
}

// fp32 feed-forward module on the bf16 matrix cores (gemm_x6.hip): t1 = LN(x) is in place;
// split it into planes, w_1 + activation straight into the plane image of the hidden
// tensor, w_2 as K-slice partials in m->ffn_part.  Returns the slice count (0: not taken).
// Which form will ffn_x6_try run for this module on the current batch?  0 = none (the caller's
// v_mfma_f32 paths), 1 = the fused kernel (ffn_x6f.hip, d_model 256), 2 = the six-product GEMM
// pair with the hidden tensor as a plane image.  Forms 1 and 2 (without tune().x6_af32) read
// LN(x) as an X3 plane image: the producers of LN(x) ask before they decide to write that image
// instead of fp32 rows (t1_image_for) -- ONE predicate, so producer and consumer cannot disagree.
int ffn_x6_route(wn_model* m, const Linear& w1, const Linear& w2, int act) {
  const int d = m->cfg.d_model, M = m->rows, F = w1.out;
  if (t_gemm_prec != PREC_F32 || tune().gemm_x6 == 0 || !m->x6_at || w1.out != w2.in ||
      !(d == 256 || d == 512) || F % 16 != 0)
    return 0;
  // batches under 512 rows: only the fused kernel (the tile-GEMM pair is all prologue and
  // epilogue there; 2 = tests force it)
  const bool small = M < 512 && tune().gemm_x6 != 2;
  const bool fused_ok = tune().ffn_x6f != 0 && tune().x6_af32 == 0 && m->x6p_at &&
                        ffn_x6f_supported(M, d, F, act) && !(small && tune().ffn_x6f == 3);
  if (m->x6_at->count(w1.w) == 0 || m->x6_at->count(w2.w) == 0) return 0;
  const bool fused = fused_ok && m->x6p_at->count(w2.w) != 0;
  // small batches never take the tile-GEMM pair: the fused kernel or the v_mfma_f32 paths
  if (small) return fused ? 1 : 0;
  return fused ? 1 : 2;
}

// the image buffer of t1 = LN(x) when the next feed-forward module will take it, else null
void* t1_image_for(wn_model* m, const Linear& w1, const Linear& w2, int act) {
  if (tune().ffn_ximg == 0) return nullptr;
  const int r = ffn_x6_route(m, w1, w2, act);
  if (r == 0 || (r == 2 && tune().x6_af32 != 0)) return nullptr;
  if (m->t1_img.ensure(x6_bytes(m->rows, m->cfg.d_model)) != 0) return nullptr;
  return m->t1_img.p;
}

int ffn_x6_try(wn_model* m, const Linear& w1, const Linear& w2, int act, hipStream_t s) {
  const int d = m->cfg.d_model, M = m->rows, F = w1.out;
  const bool ximg = m->t1_img_ok;      // t1 exists ONLY as its plane image
  m->t1_img_ok = false;
  // (d: the widths ffn_reduce_ln takes)
  const int route = ffn_x6_route(m, w1, w2, act);
  if (route == 0) {
    if (ximg) {
      set_error("ffn_x6_try: LN(x) was left as a plane image but no six-product form takes it");
      return -1;
    }
    return 0;
  }
  auto i1 = m->x6_at->find(w1.w), i2 = m->x6_at->find(w2.w);
  static thread_local int tick = 0;
  if (route == 1) {
    // hidden tensor on chip (ffn_x6f.hip)
    auto ip = m->x6p_at->find(w2.w);
    {
      FfnX6Args a;
      a.S = ffn_x6f_split(M, F);
      if (m->ffn_part.ensure((size_t)a.S * M * d * sizeof(float)) != 0) return -1;
      a.X = m->t1.as<float>(); a.ldx = d; a.W13 = i1->second; a.W2p = ip->second; a.b1 = w1.b;
      if (ximg) { a.X3 = m->t1_img.p; a.X = nullptr; }
      a.P = m->ffn_part.as<float>(); a.M = M; a.D = d; a.F = F; a.act = act;
      const bool br = m->prof_on && (tick++ % m->prof_stride) == 0;
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
  // plane images (x6_split of t1, w_1 writes the hidden planes); tune().x6_af32 (A/B knob): the A
  // operands stay plain fp32 (t1, the hidden tensor in hbuf) and are split in registers
  const bool af32 = tune().x6_af32 != 0 && (int64_t)M * F * 4 < ((int64_t)1 << 31);
  X6Args g1;
  g1.B3 = i1->second; g1.M = M; g1.N = F; g1.K = d; g1.bias = w1.b; g1.act = act;
  if (af32) {
    if (ximg) {
      set_error("ffn_x6_try: LN(x) was left as a plane image, x6_af32 wants fp32 rows");
      return -1;
    }
    if (m->hbuf.ensure((size_t)M * F * sizeof(float)) != 0) return -1;
    g1.A = m->t1.as<float>(); g1.lda = d; g1.a_bytes = (int64_t)M * d * 4;
    g1.epi = 0; g1.C = m->hbuf.as<float>(); g1.ldc = F;
  } else if (ximg) {
    // the producer of LN(x) already wrote its plane image (round 5): no x6_split launch
    if (m->x6_h.ensure(x6_bytes(M, F)) != 0) return -1;
    g1.A3 = m->t1_img.as<char>(); g1.epi = 2; g1.C3 = m->x6_h.as<char>();
  } else {
    if (m->x6_a.ensure(x6_bytes(M, d)) != 0 || m->x6_h.ensure(x6_bytes(M, F)) != 0) return -1;
    if (x6_split(m->t1.as<float>(), M, d, d, m->x6_a.as<char>(), s) != 0) return -1;
    g1.A3 = m->x6_a.as<char>(); g1.epi = 2; g1.C3 = m->x6_h.as<char>();
  }
  const bool bracket = m->prof_on && (tick++ % m->prof_stride) == 0;
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
  if (gemm_x6(g2, s) != 0) return -1;
  m->prof_split = S;
  return S;
}

// The six-product GEMM pair of a feed-forward module on ANY rows (the decoders' ReLU modules over
// the R hypothesis rows of a rescoring pass, decoder_layer.py:140-147): A = LN(x) [M][d] is split
// into planes, w_1 + activation writes the plane image of the hidden tensor from its epilogue
// (no fp32 hidden tensor, no separate split pass over it: 10 B per hidden element of HBM traffic
// less than linear() + linear()), w_2 leaves K-slice partials in m->ffn_part for ffn_reduce_ln.
// Returns the slice count, 0 if the shape stays on linear(), < 0 on error.
int ffn_x6_pair(wn_model* m, const Linear& w1, const Linear& w2, int act, const float* A, int M,
                hipStream_t s) {
  const int d = w1.in, F = w1.out;
  if (t_gemm_prec != PREC_F32 || tune().gemm_x6 == 0 || tune().x6_linear == 0 || !m->x6_at ||
      w1.out != w2.in || w2.out != d || !(d == 256 || d == 512) || F % 16 != 0 || M < 512 ||
      2.0 * M * (double)F * d < 1e8 * g_x6_linear_min)
    return 0;
  auto i1 = m->x6_at->find(w1.w), i2 = m->x6_at->find(w2.w);
  if (i1 == m->x6_at->end() || i2 == m->x6_at->end()) return 0;
  const int S = ffn_x6_split(M, F);
  if (m->ffn_part.ensure((size_t)S * M * d * sizeof(float)) != 0 ||
      m->x6_a.ensure(x6_bytes(M, d)) != 0 || m->x6_h.ensure(x6_bytes(M, F)) != 0)
    return -1;
  if (x6_split(A, M, d, d, m->x6_a.as<char>(), s) != 0) return -1;
  X6Args g1;
  g1.A3 = m->x6_a.as<char>(); g1.B3 = i1->second; g1.M = M; g1.N = F; g1.K = d; g1.bias = w1.b;
  g1.act = act; g1.epi = 2; g1.C3 = m->x6_h.as<char>();
  if (gemm_x6(g1, s) != 0) return -1;
  X6Args g2;
  g2.A3 = m->x6_h.as<char>(); g2.B3 = i2->second; g2.M = M; g2.N = d; g2.K = F;
  g2.epi = 1; g2.ksplit = S; g2.C = m->ffn_part.as<float>();
  if (gemm_x6(g2, s) != 0) return -1;
  return S;
}

// fp32 fused feed-forward module (ffn_fused.hip): t1 = LN(x) is in place; leaves the
// hidden-slice partials in m->ffn_part and returns S (0: shape not taken, caller runs
// the two-GEMM path).  Every 6th launch is bracketed for the roofline (wn_profile_*).
int ffn_fused_try(wn_model* m, const Linear& w1, const Linear& w2, int act, hipStream_t s) {
  const wn_config& c = m->cfg;
  const int d = c.d_model, M = m->rows;
  if (const int s6 = ffn_x6_try(m, w1, w2, act, s)) return s6;
  if (t_gemm_prec != PREC_F32 || tune().ffn_fused == 0 || w1.out != w2.in ||
      !ffn_fused_supported(M, d, w1.out, act))
    return 0;
  FfnArgs a;
  a.X = m->t1.as<float>(); a.W1 = w1.w; a.b1 = w1.b; a.W2 = w2.w;
  a.M = M; a.D = d; a.F = w1.out; a.S = ffn_fused_split(M, d, w1.out); a.act = act;
  if (m->ffn_part.ensure((size_t)a.S * M * d * sizeof(float)) != 0) return -1;
  a.P = m->ffn_part.as<float>();
  const bool bracket = m->prof_on && (m->prof_seq++ % m->prof_stride) == 0;
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
  if (m->kv_ready) {     // a prefetch nobody consumed may still be reading m->enc
    (void)hipStreamWaitEvent(s, m->side.e1, 0);
    m->kv_ready = false;
  }
  std::vector<int> row_utt(std::max(rows, 1), -1);
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < len[b]; ++t) row_utt[off[b] + t] = b;
  // Block list of the six-product self attention (attention_x6.hip, 64 queries per block),
  // appended to the row_utt upload: (sequence << 16 | head << 8 | query block), the blocks with
  // two live 32-query groups first, the "light" last blocks of sequences with an odd number of
  // query tiles behind them, longer sequences first inside each class.  The dispatcher deals
  // blocks to the CUs in this order, so the ones a CU gets on top of its two full blocks are
  // the cheap ones (round 6: 531 equal-looking blocks on 256 CUs -- the CUs with three set the
  // kernel's time).
  m->attn_blk_off = (int)((row_utt.size() + 15) / 16 * 16);
  m->attn_n_blk = 0;
  {
    const int H = m->cfg.n_heads;
    std::vector<int> order(B);
    for (int b = 0; b < B; ++b) order[b] = b;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return len[x] > len[y]; });
    std::vector<int> tab;
    if (B < 65536 && H < 256)
      for (int light = 0; light < 2; ++light)
        for (int b : order) {
          const int nqb = (len[b] + 63) / 64;
          for (int qb = 0; qb < nqb && qb < 256; ++qb) {
            const bool is_light = len[b] - qb * 64 <= 32;
            if ((int)is_light != light) continue;
            for (int h = 0; h < H; ++h) tab.push_back((b << 16) | (h << 8) | qb);
          }
        }
    row_utt.resize(m->attn_blk_off + tab.size(), -1);
    std::copy(tab.begin(), tab.end(), row_utt.begin() + m->attn_blk_off);
    m->attn_n_blk = (int)tab.size();
  }
  WN_TRY(m->stage.begin((size_t)(row_utt.size() + 4 * B + 64) * sizeof(int) + 1024));
  WN_TRY(upload_desc(m, m->d_off, off, s));
  WN_TRY(upload_desc(m, m->d_len, len, s));
  WN_TRY(upload_desc(m, m->d_row_utt, row_utt, s));
  m->ctc_valid = false;
  return 0;
}

// GlobalCMVN + Conv2dSubsampling4 + RelPositionalEncoding scale for a padded
// Conv2dSubsampling's out Linear(d * F2 -> d) * sqrt(d) (subsampling.py:225-226, embedding.py:144)
// on the six-product GEMM (round 3).  K = 4864 / 9728 and N = d give 31 row tiles at config 2: the
// K dimension is cut into S slices (tiles x S fills the CUs once), the fp32 rows of conv2's output
// are split into planes in registers (no plane image of the 154-MB tensor), the slice partials
// are added by ffn_reduce_ln together with the bias, the scale and layer 0's norm_ff_macaron
// (x starts as zeros: 0 + alpha (sum + b) is alpha (sum + b) exactly).  191 us on v_mfma_f32
// (102 TF, r05e) before.  Returns 1 if it ran, 0 if the shape stays on linear().
int sub_out_linear(wn_model* m, int M, int F2, hipStream_t s) {
  const wn_config& c = m->cfg;
  const int d = c.d_model, K = F2 * d;
  if (t_gemm_prec != PREC_F32 || tune().gemm_x6 == 0 || !m->x6_at || m->layers.empty() ||
      (d != 256 && d != 512) || K % 16 != 0 || M < 512 || (int64_t)M * K * 4 >= ((int64_t)1 << 31) ||
      bf16_store_active())
    return 0;
  auto it = m->x6_at->find(m->sub_out.w);
  if (it == m->x6_at->end()) return 0;
  const int bm = 256;
  const int tiles = cdiv(M, bm) * cdiv(d, 256), nkb = K / 16;
  int S = 0;
  for (int t = std::min(16, 256 / std::max(tiles, 1)); t >= 2; --t)
    if (nkb % t == 0) { S = t; break; }
  if (S < 2) return 0;
  WN_TRY(m->ffn_part.ensure((size_t)S * M * d * sizeof(float)));
  X6Args g;
  g.A = m->c2.as<float>(); g.lda = K; g.a_bytes = (int64_t)M * K * 4;
  g.B3 = it->second; g.M = M; g.N = d; g.K = K; g.epi = 1; g.ksplit = S; g.bm = bm;
  g.C = m->ffn_part.as<float>();
  WN_TRY(gemm_x6(g, s));
  WN_HIP(hipMemsetAsync(m->x.p, 0, (size_t)M * d * sizeof(float), s));
  const EncLayer& L0 = m->layers[0];
  WN_TRY(ffn_reduce_ln(m->x.as<float>(), m->ffn_part.as<float>(), S, m->sub_out.b, sqrtf((float)d),
                       L0.norm_ff_mac.w, L0.norm_ff_mac.b, nullptr, nullptr, m->t1.as<float>(), M,
                       d, c.norm_eps, 0, s));
  m->ln0_done = true;
  return 1;
}

// (B, T, F) feature batch: sets the row layout and leaves x = embed(xs) in m->x
// (encoder.py:155-157, subsampling.py:203-228, embedding.py:134-147).  `pos0` is
// the position of the first output frame (streaming offset).
// consume the handle's encode gate (wn_model_set_encode_gate): everything queued on `s` after
// this point waits for the event; idempotent
int encode_gate_wait(wn_model* m, hipStream_t s) {
  if (m->enc_gate) {
    hipEvent_t e = m->enc_gate;
    m->enc_gate = nullptr;
    WN_HIP(hipStreamWaitEvent(s, e, 0));
  }
  return 0;
}

int subsample_conv2d4(wn_model* m, const float* feats_dev,
                      const int32_t* feat_lens_host, int B, int T,
                      int32_t* enc_lens_host, int pos0, hipStream_t s) {
  const wn_config& c = m->cfg;
  const int d = c.d_model, F1 = m->F1(), F2 = m->F2();
  const int Tp = ((T - 1) / 2 - 1) / 2;
  m->ln0_done = false;
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
    // encode gate, position 0 (the default): the wait for the previous decode sits BEHIND this
    // call's descriptor uploads -- five small host -> device copies, ~22 us of copy kernels plus
    // their launch gaps that otherwise stand in the encoder chain (two-stream timeline r13b) --
    // and in front of conv1
    if (tune().enc_gate_pos == 0) WN_TRY(encode_gate_wait(m, s));
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
    if (t_gemm_prec == PREC_F32 && tune().gemm_x6 != 0 && m->x6_at && d % 32 == 0 &&
        F1 <= 64 &&
        (M * F2 >= 4096 || tune().gemm_x6 == 2)) {
      auto it = m->x6_at->find(m->conv2.w);
      if (it != m->x6_at->end()) w6 = it->second;
    }
    if (w6 && (tune().x6_af32 != 0 || tune().x6_conv_af32 != 0) &&
        (int64_t)M1 * F1 * d * 4 < ((int64_t)1 << 31)) {
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
      g.a_pix = pix; g.conv_kbc = d / 16;
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
      // the front end (HBM-bound: 10 MB of features -> the ~1-GB plane image) may run beside
      // the previous decode's matrix-bound encoder; conv2 and everything behind it may not
      WN_TRY(encode_gate_wait(m, s));
      X6Args g;
      g.A3 = m->c1.as<char>(); g.B3 = w6; g.M = M * F2; g.N = d; g.K = 9 * d;
      g.epi = 0; g.bias = m->conv2.b; g.act = ACT_RELU; g.C = m->c2.as<float>(); g.ldc = d;
      g.a_pix = pix; g.a_tiles = tiles; g.conv_kbc = d / 16;
      g.conv_taps = 9;
      {   // scratch for the K-slice partials of the last, partial round of tiles (gemm_x6.hip)
        const int ncu = 256;   // (as in gemm_x6())
        const int t256 = cdiv(M * F2, 256), rem = t256 - t256 / ncu * ncu;
        if (d <= 256 && t256 >= ncu && rem > 0 && rem <= ncu / 2) {
          const size_t need = (size_t)4 * ((size_t)M * F2 - (size_t)(t256 - rem) * 256) * d *
                              sizeof(float);
          WN_TRY(m->ffn_part.ensure(need));
          g.part = m->ffn_part.as<float>(); g.part_bytes = m->ffn_part.cap;
        }
      }
      const int ne = (F1 + 1) / 2;
      for (int ky = 0; ky < 3; ++ky) {
        g.tap_delta[ky * 3 + 0] = ky * F1;            // f1 = 2 f2     (even, position f2)
        g.tap_delta[ky * 3 + 1] = ky * F1 + ne;       // f1 = 2 f2 + 1 (odd, position ne + f2)
        g.tap_delta[ky * 3 + 2] = ky * F1 + 1;        // f1 = 2 f2 + 2 (even, position f2 + 1)
      }
      WN_TRY(gemm_x6(g, s));
      const int r = sub_out_linear(m, M, F2, s);
      if (r < 0) return r;
      if (r == 0)
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
  m->t1_img_ok = false;
  for (int li = 0; li < n_run; ++li) {
    const EncLayer& L = m->layers[li];
    // x += 0.5 * FFN_macaron(LN(x))                 encoder_layer.py:220-228
    // (for li > 0 the previous layer's tail already left LN(x) in t1)
    // fp32: fused FFN (hidden tensor stays on chip), its partial reduction carries the
    // residual add and the NEXT LayerNorm (norm_mha)
    int fS = 0;
    if (!h16 && t_gemm_prec == PREC_F32) {
      if (li == 0 && !m->ln0_done) WN_TRY(ln(L.norm_ff_mac, x, t1, M, d, eps, s));
      m->ln0_done = false;
      fS = ffn_fused_try(m, L.ffm1, L.ffm2, ACT_SILU, s);
      if (fS < 0) return -2;
    }
    // the QKV projection on the row-block kernel: it can form LN(x + 0.5 FFN) itself from the
    // slice partials (gemm_x6r.hip / gemm_x6r512.hip, PRO) -- no ffn_reduce_ln launch, no t1 round trip
    const void* qkv_w6 = nullptr;
    if (!h16 && t_gemm_prec == PREC_F32 && tune().x6r != 0 && tune().gemm_x6 != 0 && t_x6 && M >= 512 &&
        gemm_x6r_supported(M, 3 * d, L.qkv.in, 0)) {
      auto it = t_x6->find(L.qkv.w);
      if (it != t_x6->end()) qkv_w6 = it->second;
    }
    const bool pro = fS > 0 && qkv_w6 && tune().x6r_pro != 0 && d == 256;   // (d = 512: measured slower)
    if (fS > 0) {
      if (!pro)
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
    // The six-product attention over key tiles aligned to the global 32-row blocks (tune
    // attn_x6_galign): decided HERE, in front of the QKV projection, because with = 2 that
    // projection writes the tile images itself (epi 4) and leaves no K / V rows behind
    const bool ax6 = !h16 && t_gemm_prec == PREC_F32 && tune().attn_x6 != 0 &&
                     tune().attn_fold == 1 && mask_mode == 0 && M >= 512 && max_len >= 128 &&
                     L.pos_tab && L.bias_u && L.bias_v;
    bool ax6_img = false;
    if (ax6 && m->attn_img.ensure(attention_x6_image_bytes(M, m->B, c.n_heads)) == 0)
      ax6_img = true;
    const std::pair<const void*, const float*>* qkv_q = nullptr;
    if (ax6_img && pro && tune().attn_x6_galign == 2 && m->x6q_at) {
      auto it = m->x6q_at->find(L.qkv.w);
      if (it != m->x6q_at->end()) qkv_q = &it->second;
    }
    bool qkv_done = false;
    if (qkv_w6) {
      X6RArgs g;
      g.A = t1; g.lda = d; g.K = d; g.W3 = qkv_w6; g.bias = L.qkv.b; g.M = M; g.N = 3 * d;
      g.epi = 0; g.C = qkv; g.ldc = 3 * d;
      if (qkv_q) {
        g.epi = 4; g.W3 = qkv_q->first; g.bias = qkv_q->second;
        g.at_img = m->attn_img.p; g.at_P = L.pos_tab; g.at_ldp = d;
        g.at_u = L.bias_u; g.at_v = L.bias_v;
        g.at_row_utt = m->d_row_utt.as<int>(); g.at_off = m->d_off.as<int>();
      }
      if (pro) {
        g.pro_P = m->ffn_part.as<float>(); g.pro_S = fS; g.pro_b2 = L.ffm2.b; g.pro_alpha = 0.5f;
        g.pro_x = x; g.ln_w = L.norm_mha.w; g.ln_b = L.norm_mha.b; g.eps = eps;
      }
      WN_TRY(gemm_x6r(g, s));
      qkv_done = true;
    }
    if (!qkv_done)
      WN_TRY(linear(L.qkv, t1, d, qkv, 3 * d, M, s, ACT_NONE, nullptr, 0, 1.0f, false,
                    h16));
    AttnArgs a;
    a.Q = qkv; a.K = qkv + d; a.V = qkv + 2 * d;
    a.ldq = a.ldk = a.ldv = 3 * d;
    a.P = L.pos_tab; a.ldp = d; a.bias_u = L.bias_u; a.bias_v = L.bias_v;
    if (!h16 && t_gemm_prec == PREC_F32 && tune().attn_fold == 2 && (d == 256 || d == 512) &&
        d == c.n_heads * 64) {
      // A/B form: the folding as a separate pass (k <- k + p in place, scalars in HBM)
      WN_TRY(m->attn_kbias.ensure((size_t)M * c.n_heads * sizeof(float)));
      WN_TRY(relpos_fold(qkv + d, 3 * d, L.pos_tab, d, L.bias_u, L.bias_v,
                         m->d_row_utt.as<int>(), m->d_off.as<int>(), nullptr,
                         m->attn_kbias.as<float>(), c.n_heads, M, d, s));
      a.kbias = m->attn_kbias.as<float>();
      a.P = nullptr; a.bias_u = a.bias_v = nullptr;
    } else if (!h16 && t_gemm_prec == PREC_F32 && tune().attn_fold != 0) {
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
    if (ax6_img) {
      a.x6_img = m->attn_img.p; a.x6_img_bytes = m->attn_img.cap; a.x6_rows = M;
      if (tune().attn_x6_galign != 0) {
        a.x6_galign = 1;
        a.row_utt = m->d_row_utt.as<int>();
      }
    } else if (a.fold && tune().attn_x6 != 0 && !h16 && t_gemm_prec == PREC_F32 && M >= 512) {
      // (chunk-masked batches with attn_x6 = 2: the sequence-aligned image)
      const size_t need = attention_x6_image_bytes(M, m->B, c.n_heads);
      if (m->attn_img.ensure(need) == 0) {
        a.x6_img = m->attn_img.p; a.x6_img_bytes = m->attn_img.cap; a.x6_rows = M;
      }
    }
    // the encoder's self attention is dispatched from the batch's block list (set_layout) --
    // both fp32 kernels decode it
    if (!h16 && t_gemm_prec == PREC_F32 && tune().attn_x6_order != 0 && m->attn_n_blk > 0) {
      a.blk_tab = m->d_row_utt.as<int>() + m->attn_blk_off;
      a.n_blk = m->attn_n_blk;
    }
    if (qkv_q) {
      // the tiles are in the image already and K / V exist nowhere else: this launch MUST be
      // the six-product kernel
      a.x6_img_ready = true;
      WN_CHECK(a.fold && attention_x6_supported(a),
               "encoder: QKV wrote the key-tile images but the six-product attention declines");
      WN_TRY(attention_x6(a, s));
    } else {
      WN_TRY(attention(a, s));
    }
    // x += out_proj(context); t1 = LN_conv(x)       encoder_layer.py:236-240
    const bool rowln = !h16 && t_gemm_prec == PREC_F32 && gemm_rowln_supported(M, d, d);
    // the same fusion as six bf16 plane products on the row-block kernels (gemm_x6r.hip: A rows
    // in registers, d = 256; gemm_x6r512.hip: A image in LDS, d = 512 -- no v_mfma_f32 row-LN
    // kernel exists at that width)
    const bool rowx = rowln || (!h16 && t_gemm_prec == PREC_F32 && d == 512);
    auto x6r_rowln = [&](const Linear& l, const float* A, const Norm& nrm,
                         const DwConvArgs* dwc = nullptr, void* y_img = nullptr) -> int {
      if (!(rowx && tune().x6r != 0 && tune().gemm_x6 != 0 && t_x6 && M >= 512 &&
            gemm_x6r_supported(M, d, l.in, 1)))
        return 1;
      auto it = t_x6->find(l.w);
      if (it == t_x6->end()) return 1;
      X6RArgs g;
      if (dwc) { g.dw = *dwc; g.dw_on = 1; }
      g.A = A; g.lda = d; g.K = d; g.W3 = it->second; g.bias = l.b; g.M = M; g.N = d; g.epi = 1;
      g.resid = x; g.ldr = d; g.alpha = 1.0f; g.x_out = x; g.ldx = d;
      g.ln_w = nrm.w; g.ln_b = nrm.b; g.eps = eps; g.y = t1; g.ldy = d;
      if (y_img) { g.y3 = y_img; g.y = nullptr; }     // LN(x) leaves as its plane image only
      if (gemm_x6r(g, s) != 0) return -1;
      m->t1_img_ok = y_img != nullptr;
      return 0;
    };
    // out-projection + residual + LN_conv chained with pointwise_conv1 + GLU in ONE launch
    // (gemm_x6r.hip epi 3): LN_conv(x) never reaches HBM
    bool pw1_done = false;
    int xr = 1;
    if (rowx && tune().x6r >= 1 && tune().x6r_chain != 0 && tune().gemm_x6 != 0 && t_x6 && M >= 512 &&
        gemm_x6r_supported(M, d, L.out.in, 3) &&
        (d == 512 || gemm_x6r_supported(M, 2 * d, L.pw1.in, 2))) {
      auto io = t_x6->find(L.out.w), ip = t_x6->find(L.pw1.w);
      if (io != t_x6->end() && ip != t_x6->end()) {
        X6RArgs g;
        g.A = t2; g.lda = d; g.K = d; g.W3 = io->second; g.bias = L.out.b; g.M = M; g.N = d; g.epi = 3;
        g.resid = x; g.ldr = d; g.alpha = 1.0f; g.x_out = x; g.ldx = d;
        g.ln_w = L.norm_conv.w; g.ln_b = L.norm_conv.b; g.eps = eps; g.y = nullptr; g.ldy = d;
        g.W3b = ip->second; g.bias2 = L.pw1.b; g.C = t2; g.ldc = d;   // (C aliases A: a block reads its rows first)
        if (gemm_x6r(g, s) != 0) return -2;
        xr = 0; pw1_done = true;
      }
    }
    if (xr != 0) xr = x6r_rowln(L.out, t2, L.norm_conv);
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
    // pointwise_conv1 + GLU                        convolution.py:115-118
    if (!pw1_done && rowx && tune().x6r != 0 && tune().gemm_x6 != 0 && t_x6 && M >= 512 &&
        gemm_x6r_supported(M, 2 * d, L.pw1.in, 2)) {
      auto it = t_x6->find(L.pw1.w);
      if (it != t_x6->end()) {
        X6RArgs g;
        g.A = t1; g.lda = d; g.K = d; g.W3 = it->second; g.bias = L.pw1.b; g.M = M; g.N = 2 * d;
        g.epi = 2; g.C = t2; g.ldc = d;
        WN_TRY(gemm_x6r(g, s));
        pw1_done = true;
      }
    }
    if (!pw1_done)
      WN_TRY(linear(L.pw1, t1, d, t2, d, M, s, ACT_NONE, nullptr, 0, 1.0f, true, h16));
    DwConvArgs dw;
    dw.x = t2; dw.ldx = d; dw.wt = L.dw_wt; dw.bias = L.dw_b; dw.cpad = L.cpad;
    dw.ln_w = L.conv_norm.w; dw.ln_b = L.conv_norm.b; dw.norm_mode = c.cnn_norm;
    dw.y = t1; dw.ldy = d;
    dw.row_utt = m->d_row_utt.as<int>(); dw.off = m->d_off.as<int>();
    dw.len = m->d_len.as<int>();
    dw.M = M; dw.D = d; dw.K = c.cnn_kernel; dw.causal = c.causal;
    dw.t_max = m->Tp; dw.eps = 1e-5f;
    // x += pointwise_conv2(.); t1 = LN_ff(x)        encoder_layer.py:251-255
    // d = 256 on the row-block kernel: the depthwise conv + norm + SiLU is its prologue
    // (gemm_x6r.hip DWC) -- no launch, no round trip of the conv module's middle tensor
    const bool dwc = tune().x6r_dwc != 0 && d == 256 && rowx && tune().x6r != 0 && tune().gemm_x6 != 0 && t_x6 &&
                     M >= 512 && t_x6->count(L.pw2.w) != 0 && gemm_x6r_supported(M, d, L.pw2.in, 1);
    if (!dwc) WN_TRY(dwconv_ln_silu(dw, s));
    // (the feed-forward module behind it takes LN_ff(x) as a plane image where it runs fused:
    // asked BEFORE the launch, the row-block kernel is the only producer that can write one)
    void* ff_img = nullptr;
    if (!h16 && rowx && tune().x6r != 0 && tune().gemm_x6 != 0 && t_x6 && M >= 512 &&
        gemm_x6r_supported(M, d, L.pw2.in, 1) && t_x6->count(L.pw2.w) != 0)
      ff_img = t1_image_for(m, L.ff1, L.ff2, ACT_SILU);
    xr = x6r_rowln(L.pw2, t1, L.norm_ff, dwc ? &dw : nullptr, ff_img);
    if (xr < 0) return -2;
    const bool ln_ff_done = xr == 0 || rowln;       // t1 = LN_ff(x) came out of the GEMM's epilogue
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
      if (!ln_ff_done) WN_TRY(ln(L.norm_ff, x, t1, M, d, eps, s));
      fS = ffn_fused_try(m, L.ff1, L.ff2, ACT_SILU, s);
      if (fS < 0) return -2;
    }
    if (fS > 0) {
      // partial reduction + residual + norm_final (+ the next layer's norm_ff_macaron)
      if (li + 1 < n_run) {
        const EncLayer& Ln = m->layers[li + 1];
        void* mac_img = t1_image_for(m, Ln.ffm1, Ln.ffm2, ACT_SILU);
        if (mac_img) {
          WN_TRY(ffn_reduce_ln_img(x, m->ffn_part.as<float>(), fS, L.ff2.b, 0.5f, L.norm_final.w,
                                   L.norm_final.b, Ln.norm_ff_mac.w, Ln.norm_ff_mac.b, mac_img,
                                   M, d, eps, s));
          m->t1_img_ok = true;
        } else {
          WN_TRY(ffn_reduce_ln(x, m->ffn_part.as<float>(), fS, L.ff2.b, 0.5f, L.norm_final.w,
                               L.norm_final.b, Ln.norm_ff_mac.w, Ln.norm_ff_mac.b, t1, M, d, eps,
                               1, s));
        }
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
    const bool q16 = h16 && tune().qkv_bf16 != 0;
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
  if (m->kv_ready) {     // a prefetch nobody consumed may still be reading m->enc
    (void)hipStreamWaitEvent(s, m->side.e1, 0);
    m->kv_ready = false;
  }
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
      m->attn_n_blk = 0;      // (no block list behind this row_utt)
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

