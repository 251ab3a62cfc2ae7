// The C ABI of libwenet_amd (include/wenet_amd.h): handle life cycle and weight ingestion, one
// entry point per reference function of the decode path, the searches' and the rescoring
// decoder's launch sequences, the operator hooks the tests use.  State: model_state.h; the
// encoder engine it calls into: model.hip.
#include "model_state.h"

namespace {

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

const char* wn_last_error(void) { return last_error_cstr(); }
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
  m->tune_ovr = src->tune_ovr;
  // weights, projected position tables and fbank tables are read-only: share
  m->weights = src->weights;
  m->n_weight_elems = src->n_weight_elems;
  m->weights_bf16 = src->weights_bf16;
  m->weights_mx = src->weights_mx; m->mx_at = src->mx_at; m->fp8_ffn = src->fp8_ffn;
  m->weights_x6 = src->weights_x6; m->x6_at = src->x6_at;
  m->weights_x6p = src->weights_x6p; m->x6p_at = src->x6p_at; m->bias4_buf = src->bias4_buf; m->bias4 = src->bias4;
  m->weights_x6q = src->weights_x6q; m->x6q_at = src->x6q_at;
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

int32_t wn_batch_size(const wn_model* m) { return m ? m->B : -1; }

int wn_model_set_encode_gate(wn_model* m, void* event) {
  WN_CHECK(m, "wn_model_set_encode_gate: null handle");
  m->enc_gate = (hipEvent_t)event;
  return 0;
}

int wn_profile_enable(wn_model* m, int32_t on) {
  WN_CHECK(m, "wn_profile_enable: null model");
  m->prof_on = on != 0;
  m->prof_stride = on > 1 ? (unsigned)on : 6u;   // on = 1: every 6th launch; on = N > 1: every N-th
  m->prof_used = 0;
  m->prof_flops = 0.0;
  return 0;
}

const char* wn_profile_kernel_name(const wn_model* m) {
  return m ? m->prof_kernel : "";
}

int32_t wn_profile_ffn_split(const wn_model* m) { return m ? m->prof_split : 0; }

int wn_profile_gemm_clocks(uint64_t* out64) {
  WN_CHECK(out64, "wn_profile_gemm_clocks: null output");
  if (wn::tune().x6_probe == 8)   // the row-block kernel's phase stamps (tools/x6r_clocks.py)
    return wn::gemm_x6r_clocks(reinterpret_cast<unsigned long long*>(out64));
  if (wn::tune().lp_probe & 4)   // the pipelined bf16 / MXFP8 kernel stamped last (tools/lp_clocks.py)
    return wn::gemm_lp_clocks(reinterpret_cast<unsigned long long*>(out64));
  return wn::gemm_x6_clocks(reinterpret_cast<unsigned long long*>(out64));
}

int wn_profile_ffn_clocks(uint64_t* out64) {
  WN_CHECK(out64, "wn_profile_ffn_clocks: null output");
  return wn::ffn_x6f_clocks(reinterpret_cast<unsigned long long*>(out64));
}

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
  int* f = tune_field(g_tune_default, k);
  if (!f) { set_error("wn_tune_set: unknown key " + k); return -1; }
  WN_CHECK(value != TUNE_INHERIT, "wn_tune_set: INT32_MIN is the per-handle 'inherit' marker");
  if (tune_check(k, value, "wn_tune_set") != 0) return -1;
  *f = value;
  return 0;
}

int wn_model_tune_set(wn_model* m, const char* key, int32_t value) {
  WN_CHECK(m && key, "wn_model_tune_set: null argument");
  WN_ENTER(m);
  const std::string k(key);
  int* f = tune_field(m->tune_ovr, k);
  if (!f) { set_error("wn_model_tune_set: unknown key " + k); return -1; }
  if (tune_check(k, value, "wn_model_tune_set") != 0) return -1;
  *f = value;
  return 0;
}

int wn_tune_get(const wn_model* m, const char* key, int32_t* value) {
  WN_CHECK(key && value, "wn_tune_get: null argument");
  const std::string k(key);
  Tune eff = g_tune_default;
  if (m) tune_resolve(m->tune_ovr, &eff);
  const int* f = tune_field(eff, k);
  if (!f) { set_error("wn_tune_get: unknown key " + k); return -1; }
  *value = *f;
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
  // the encode gate is one-shot: whatever way this call ends (an argument check included), it
  // does not stay on the handle for a later call to wait on an event that may be gone by then
  struct GateDrop { wn_model* m; ~GateDrop() { m->enc_gate = nullptr; } } gate_drop{m};
  m->pb_valid = false;
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
    WN_TRY(encode_gate_wait(m, (hipStream_t)stream));
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
  WN_TRY(encode_gate_wait(m, s));     // (paths that did not consume the gate behind conv1)
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
  m->pb_valid = false;
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
  m->pb_valid = false;
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
    // logits rows at a pitch of V rounded up to 32 floats: whole 128-byte lines per row, so
    // that the GEMM's column tiles (multiples of 128 columns) never share a line -- at a pitch
    // of V rounded up to 4 (round 5) the tile edges fell inside lines written by two blocks on
    // two XCDs, and the counters showed the read-modify-write: 140 MB READ + 168 MB written
    // by a GEMM whose operands are 19 MB and whose result is 134 MB (r13b)
    const int V4 = (V + 31) / 32 * 32;
    ldv = V4;
    WN_TRY(m->logits.ensure((size_t)M * V4 * sizeof(float)));
    WN_TRY(m->topk_val.ensure((size_t)M * k * sizeof(float)));
    WN_TRY(m->topk_idx.ensure((size_t)M * k * sizeof(int)));
    WN_TRY(vocab_linear(m, m->ctc, m->enc.as<float>(), c.d_model, m->logits.as<float>(), V4, M,
                        s));
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
  m->pb_valid = false;
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
  WN_HIP(stream_wait(s));
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
  WN_HIP(stream_wait(s));
  int T = 0;
  for (int b = 0; b < B; ++b) { n_keep_host[b] = keep[b]; T = std::max(T, keep[b]); }
  *t_out = T;
  if (T == 0) {
    // a batch of silence: the reference fails here (pad_sequence of empty selections,
    // asr_model.py:165-172).  One silent batch must not end a long recognize.py run: the
    // layout and the encoder output stay as they are (rescoring then attends to the
    // unfiltered frames) and the caller is told through *t_out == 0.
    static bool warned = false;
    if (!warned) {
      fprintf(stderr, "[wenet_amd] filter_blank_embedding: no non-blank frame in the whole "
                      "batch; the encoder output is left unfiltered\n");
      warned = true;
    }
    return 0;
  }
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
  // the results in ONE device block -> one copy into pinned memory (six staged copies into the
  // caller's pageable arrays took ~130 us of host round trips per batch, r05f trace)
  const size_t o_sc = 0, o_nh = o_sc + nb * sizeof(double), o_len = o_nh + (size_t)B * sizeof(int),
               o_tlen = o_len + nb * sizeof(int), o_tok = o_tlen + nb * sizeof(int),
               o_tim = o_tok + nb * max_len * sizeof(int),
               o_end = o_tim + nb * max_len * sizeof(int);
  WN_TRY(m->pb_out.ensure(o_end));
  WN_TRY(m->pb_host.ensure(o_end));
  char* ob = m->pb_out.as<char>();
  PrefixBeamArgs a;
  a.topk_val = m->topk_val.as<float>(); a.topk_idx = m->topk_idx.as<int>();
  a.k = m->ctc_k; a.off = m->d_off.as<int>(); a.len = m->d_len.as<int>();
  a.B = B; a.beam = beam; a.blank = blank_id; a.max_len = max_len;
  a.pool = m->pb_pool.as<int>(); a.pool_stride = pool;
  a.n_hyps = reinterpret_cast<int*>(ob + o_nh); a.hyp_lens = reinterpret_cast<int*>(ob + o_len);
  a.hyp_tlens = reinterpret_cast<int*>(ob + o_tlen);
  a.hyp_tokens = reinterpret_cast<int*>(ob + o_tok);
  a.hyp_times = reinterpret_cast<int*>(ob + o_tim);
  a.hyp_scores = reinterpret_cast<double*>(ob + o_sc);
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
  m->pb_valid = false;
  WN_HIP(hipMemcpyAsync(m->pb_host.p, ob, o_end, hipMemcpyDeviceToHost, s));
  WN_HIP(stream_wait(s));
  m->pb_valid = true; m->pb_B = B; m->pb_beam = beam; m->pb_max_len = max_len;
  m->pb_o_sc = o_sc; m->pb_o_nh = o_nh; m->pb_o_len = o_len; m->pb_o_tok = o_tok;
  const char* hb = m->pb_host.p;
  memcpy(n_hyps_host, hb + o_nh, (size_t)B * sizeof(int));
  memcpy(hyp_lens_host, hb + o_len, nb * sizeof(int));
  memcpy(hyp_tlens_host, hb + o_tlen, nb * sizeof(int));
  memcpy(hyp_scores_host, hb + o_sc, nb * sizeof(double));
  // tokens / times: only the used corner of each [max_len] row (the caller's arrays are
  // zero-initialised; the kernel writes nothing past a hypothesis' length that anyone reads)
  const int* hl = reinterpret_cast<const int*>(hb + o_len);
  const int* htl = reinterpret_cast<const int*>(hb + o_tlen);
  for (size_t i = 0; i < nb; ++i) {
    const int nl = std::min(std::max(hl[i], 0), (int)max_len);
    const int ntl = std::min(std::max(htl[i], 0), (int)max_len);
    memcpy(hyp_tokens_host + i * max_len, hb + o_tok + i * max_len * sizeof(int), nl * sizeof(int));
    memcpy(hyp_times_host + i * max_len, hb + o_tim + i * max_len * sizeof(int), ntl * sizeof(int));
  }
  return 0;
}

// ---------------------------------------------------------------------------
namespace {
// embed + the decoder layers over a ragged batch of R token rows (n_seq
// sequences); the result stays in m->r_x.  With `mem_cache` the cross-attention
// K/V projections of the encoder output are computed once per batch and layer
// and reused by later calls (the autoregressive search calls this per step).
// Cross attention over GROUPS of sequences that share their keys (the hypotheses of one
// utterance in a rescoring pass: consecutive rows, the same encoder frames): one attention
// "sequence" per group instead of one per hypothesis -- full 64-query tiles and the K / V rows
// staged once per 64 queries instead of once per hypothesis.  Per query row the same keys in the
// same tile order: the same bits.
struct CrossGroups {
  const int* q_off; const int* q_len; const int* kv_off; const int* kv_len;
  int n_seq, max_q;
};

int decoder_layers(wn_model* m, const Decoder& D, int R, int n_seq, int max_q,
                   const int* d_tok, bool mem_cache, hipStream_t s,
                   const int* self_kvlen = nullptr, const float* kv_base = nullptr,
                   const CrossGroups* cg = nullptr) {
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
  bool ln1_done = false;      // t1 already holds this layer's norm1(x) (the previous FFN's reduce)
  for (const DecLayer& L : D.layers) {
    // causal self attention                             decoder_layer.py:100-121
    if (!ln1_done) WN_TRY(ln(L.n1, x, t1, R, d, eps, s));
    ln1_done = false;
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
    // kv_base: projected ahead of this pass (wn_rescore_prefetch)
    const float* mem = kv_base ? kv_base + (size_t)li * mem_layer
                       : mem_cache ? m->r_mem_all.as<float>() + (size_t)li * mem_layer
                                   : m->r_mem.as<float>();
    if (!kv_base && (!mem_cache || fill_cache))
      WN_TRY(linear(L.src_kv, m->enc.as<float>(), d, const_cast<float*>(mem), 2 * d, Menc, s));
    AttnArgs cx;
    cx.Q = t2; cx.ldq = d; cx.K = mem; cx.V = mem + d; cx.ldk = cx.ldv = 2 * d;
    cx.O = t1; cx.ldo = d;
    cx.q_off = m->r_qoff.as<int>(); cx.q_len = m->r_qlen.as<int>();
    cx.kv_off = m->r_kvoff.as<int>(); cx.kv_len = m->r_kvlen.as<int>();
    cx.n_seq = n_seq; cx.n_heads = c.dec_heads; cx.max_q_len = max_q;
    if (cg) {
      cx.q_off = cg->q_off; cx.q_len = cg->q_len; cx.kv_off = cg->kv_off; cx.kv_len = cg->kv_len;
      cx.n_seq = cg->n_seq; cx.max_q_len = cg->max_q;
    }
    cx.mask_mode = 0; cx.scale = 0.125f;
    WN_TRY(attention(cx, s));
    WN_TRY(linear(L.src_out, t1, d, x, d, R, s, ACT_NONE, x, d));
    // FFN (ReLU)                                         decoder_layer.py:140-147
    WN_TRY(ln(L.n3, x, t1, R, d, eps, s));
    // large batches (a rescoring pass): the six-product GEMM pair with the hidden tensor as a
    // plane image; its reduce adds b_2 and the residual and applies the NEXT LayerNorm (the
    // next layer's norm1, or after_norm behind the last layer: the callers' own after_norm
    // call then recomputes the same rows)
    const int fS = ffn_x6_pair(m, L.ff1, L.ff2, ACT_RELU, t1, R, s);
    if (fS < 0) return -2;
    if (fS > 0) {
      const bool last = (size_t)li + 1 == D.layers.size();
      const Norm& nx = last ? D.after : D.layers[li + 1].n1;
      WN_TRY(ffn_reduce_ln(x, m->ffn_part.as<float>(), fS, L.ff2.b, 1.0f, nx.w, nx.b, nullptr,
                           nullptr, t1, R, d, eps, 0, s));
      ln1_done = !last;
    } else {
      WN_TRY(linear(L.ff1, t1, d, hb, c.dec_ffn_dim, R, s, ACT_RELU));
      WN_TRY(linear(L.ff2, hb, c.dec_ffn_dim, x, d, R, s, ACT_NONE, x, d));
    }
    ++li;
  }
  if (fill_cache) m->mem_cache_valid = true;
  return 0;
}

int run_decoder(wn_model* m, const Decoder& D, int R, int n_seq, int max_q,
                const int* d_tok, const int* d_tgt, float* out_dev,
                hipStream_t s, const float* kv_base = nullptr, const CrossGroups* cg = nullptr) {
  const wn_config& c = m->cfg;
  const int d = c.d_model, V = c.vocab;
  WN_TRY(decoder_layers(m, D, R, n_seq, max_q, d_tok, false, s, nullptr, kv_base, cg));
  float* t1 = m->r_t1.as<float>();
  WN_TRY(ln(D.after, m->r_x.as<float>(), t1, R, d, c.norm_eps, s));
  // (the caller sized r_logits for a pitch of V rounded up to 4)
  WN_TRY(vocab_linear(m, D.out, t1, d, m->r_logits.as<float>(), (V + 3) / 4 * 4, R, s));
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
  WN_HIP(stream_wait(s));
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
  WN_HIP(stream_wait(s));
  for (int r = 0; r < R; ++r) {
    l2r_logp_host[out_index[r]] = hl[r];
    r2l_logp_host[out_index[r]] = hr[r];
  }
  return 0;
}

// ---------------------------------------------------------------------------
namespace {
// Decoder input / target rows of one hypothesis sequence q = (utterance, hypothesis slot):
// ys_in = [sos] + hyp (add_sos_eos, common.py:113-155), the reversed sequence for the
// right-to-left decoder (asr_model.py:491-536), and what each position is scored on
// (search.py:431-449: hyp[j] at position j, eos at position L; r_decoder_out[L-1-j] scores
// hyp[j], so the reversed rows score their own next token).  The n-best is read where the
// prefix beam search left it on the device (or where wn_rescore uploaded it).
__global__ void rescore_rows_kernel(const int* __restrict__ seq_src, const int* __restrict__ qoff,
                                    const int* __restrict__ qlen,
                                    const int* __restrict__ hyp_tokens, int max_len, int sos,
                                    int eos, int V, int* __restrict__ tok, int* __restrict__ rtok,
                                    int* __restrict__ pos, int* __restrict__ tgt,
                                    int* __restrict__ rtgt) {
  const int q = blockIdx.x;
  const int L = qlen[q] - 1, o = qoff[q];
  const int* h = hyp_tokens + (int64_t)seq_src[q] * max_len;
  for (int j = threadIdx.x; j <= L; j += blockDim.x) {
    // ids are < V by construction (top-k indices / host-checked); the clamp only keeps a
    // corrupted buffer from indexing outside the embedding table
    const int a = j == 0 ? sos : min(max(h[j - 1], 0), V - 1);
    const int r = j == 0 ? sos : min(max(h[L - j], 0), V - 1);
    tok[o + j] = a;
    rtok[o + j] = r;
    pos[o + j] = j;
    tgt[o + j] = j < L ? min(max(h[j], 0), V - 1) : eos;
    rtgt[o + j] = j < L ? min(max(h[L - 1 - j], 0), V - 1) : eos;
  }
}

struct RescoreArgs {
  const float* lp_l; const float* lp_r;      // [R] log-prob of each row's target, both decoders
  const int* seq_first;                      // [B + 1] first sequence of each utterance
  const int* seq_src;                        // [n_seq] (utt * beam + hyp slot)
  const int* qoff; const int* qlen;          // [n_seq] first row, rows (= len(hyp) + 1)
  const double* ctc_scores;                  // [B][beam] (DecodeResult.nbest_scores)
  int beam, max_len, use_r2l;
  double ctc_weight;
  float w_l, w_r;                            // fp32(1 - reverse_weight), fp32(reverse_weight)
  int* best_idx; float* best_score; double* conf; float* all_scores; double* tok_conf;
};

// The score arithmetic of attention_rescoring (search.py:424-457) for one utterance per
// wavefront, one hypothesis per lane, in the reference's dtypes and ORDER: the gathered
// log-probs are fp32 tensor elements, `score` accumulates them in fp32 left to right (the
// right-to-left decoder's from position L-1 down), Python-float operands are rounded to fp32
// where they meet the fp32 tensor (1 - reverse_weight, reverse_weight, ctc_score * ctc_weight
// -- that product itself is fp64), math.exp() is fp64.  __f*_rn: no FMA contraction.
__global__ __launch_bounds__(64) void rescore_reduce_kernel(RescoreArgs a) {
  __shared__ float s_score[64];
  __shared__ double s_conf[64];
  __shared__ int s_best;
  const int b = blockIdx.x, i = threadIdx.x;
  const int q0 = a.seq_first[b], n = a.seq_first[b + 1] - q0;
  if (i < n) {
    const int q = q0 + i, L = a.qlen[q] - 1;
    const float* l = a.lp_l + a.qoff[q];
    float score = 0.f;
    for (int j = 0; j < L; ++j) score = __fadd_rn(score, l[j]);
    score = __fadd_rn(score, l[L]);
    if (a.use_r2l) {
      const float* r = a.lp_r + a.qoff[q];
      float rs = 0.f;
      for (int j = 0; j < L; ++j) rs = __fadd_rn(rs, r[L - 1 - j]);
      rs = __fadd_rn(rs, r[L]);
      score = __fadd_rn(__fmul_rn(score, a.w_l), __fmul_rn(rs, a.w_r));
    }
    s_conf[i] = exp((double)__fdiv_rn(score, (float)(L + 1)));
    const int slot = a.seq_src[q];
    score = __fadd_rn(score, (float)(a.ctc_scores[slot] * a.ctc_weight));
    s_score[i] = score;
    a.all_scores[slot] = score;
  }
  __syncthreads();
  if (i == 0) {
    // `if score > best_score` from -inf, first maximum wins, NaN never does
    int best = 0;
    float bs = -INFINITY;
    for (int k = 0; k < n; ++k)
      if (s_score[k] > bs) { bs = s_score[k]; best = k; }
    s_best = best;
    a.best_idx[b] = n > 0 ? a.seq_src[q0 + best] - b * a.beam : 0;
    a.best_score[b] = bs;
    a.conf[b] = n > 0 ? s_conf[best] : 0.0;
  }
  __syncthreads();
  if (n <= 0) return;
  const int q = q0 + s_best, L = a.qlen[q] - 1;
  const float* l = a.lp_l + a.qoff[q];
  const float* r = a.lp_r + a.qoff[q];
  for (int j = i; j < L; j += 64) {
    double c = exp((double)l[j]);
    if (a.use_r2l) c = (c + exp((double)r[L - 1 - j])) / 2;
    a.tok_conf[(int64_t)b * a.max_len + j] = c;
  }
}
}  // namespace

int wn_rescore_prefetch(wn_model* m, int32_t use_right_decoder, void* stream) {
  WN_CHECK(m && m->B > 0 && m->enc.p, "wn_rescore_prefetch: no current batch");
  WN_ENTER(m);
  PrecisionScope prec_scope(m);
  hipStream_t s = (hipStream_t)stream;
  if (m->kv_ready) {       // an earlier prefetch of this batch: ordered behind it
    WN_HIP(hipStreamWaitEvent(s, m->side.e1, 0));
    m->kv_ready = false;
  }
  if (tune().rescore_prefetch == 0 || m->left.layers.empty() || m->rows <= 0) return 0;
  WN_HIP(hipSetDevice(m->device));
  const int d = m->cfg.d_model, Menc = m->rows;
  const bool r2l = use_right_decoder != 0 && !m->right.layers.empty();
  std::vector<const Linear*> kv;
  for (const DecLayer& L : m->left.layers) kv.push_back(&L.src_kv);
  if (r2l) for (const DecLayer& L : m->right.layers) kv.push_back(&L.src_kv);
  const size_t mem_layer = (size_t)Menc * 2 * d;
  WN_TRY(m->r_kv_all.ensure(kv.size() * mem_layer * sizeof(float)));
  WN_TRY(m->side.ensure());
  WN_HIP(hipEventRecord(m->side.e0, s));            // the encoder output is complete
  WN_HIP(hipStreamWaitEvent(m->side.st, m->side.e0, 0));
  hipStream_t ss = m->side.st;
  // all layers project the SAME rows: split the encoder output into planes once and run the
  // six-product GEMM per layer (linear() would split it once per layer) -- where linear()
  // would take that route at all
  bool x6ok = t_gemm_prec == PREC_F32 && tune().gemm_x6 != 0 && tune().x6_linear != 0 && t_x6 &&
              Menc >= 512 && d % 16 == 0 &&
              2.0 * Menc * (2.0 * d) * d >= 1e8 * 60;
  if (x6ok)
    for (const Linear* l : kv) x6ok = x6ok && t_x6->count(l->w) != 0;
  if (x6ok) {
    WN_TRY(m->r_enc3.ensure(x6_bytes(Menc, d)));
    WN_TRY(x6_split(m->enc.as<float>(), Menc, d, d, m->r_enc3.as<char>(), ss));
  }
  for (size_t i = 0; i < kv.size(); ++i) {
    float* dst = m->r_kv_all.as<float>() + i * mem_layer;
    if (x6ok) {
      X6Args x;
      x.A3 = m->r_enc3.as<char>(); x.B3 = t_x6->find(kv[i]->w)->second; x.M = Menc;
      x.N = 2 * d; x.K = d; x.epi = 0; x.bias = kv[i]->b; x.C = dst; x.ldc = 2 * d;
      WN_TRY(gemm_x6(x, ss));
    } else {
      WN_TRY(linear(*kv[i], m->enc.as<float>(), d, dst, 2 * d, Menc, ss));
    }
  }
  WN_HIP(hipEventRecord(m->side.e1, ss));
  m->kv_ready = true; m->kv_rows = Menc;
  m->kv_nl = (int)m->left.layers.size(); m->kv_nr = r2l ? (int)m->right.layers.size() : 0;
  return 0;
}

int wn_rescore(wn_model* m, int32_t beam, const int32_t* n_hyps_host,
               const int32_t* hyp_lens_host, const int32_t* hyp_tokens_host,
               const double* ctc_scores_host, int32_t max_len, double ctc_weight,
               double reverse_weight, int32_t* best_idx_host, float* best_score_host,
               double* confidence_host, double* tok_conf_host, float* all_scores_host,
               void* stream) {
  WN_CHECK(m && m->B > 0 && m->enc.p, "wn_rescore: no current batch");
  WN_ENTER(m);
  PrecisionScope prec_scope(m);
  WN_CHECK(!m->left.layers.empty(), "wn_rescore: the model has no attention decoder");
  WN_CHECK(best_idx_host && best_score_host, "wn_rescore: null output");
  hipStream_t s = (hipStream_t)stream;
  WN_HIP(hipSetDevice(m->device));
  const wn_config& c = m->cfg;
  const int B = m->B, d = c.d_model, V = c.vocab;
  // ---- where the n-best comes from -----------------------------------------------------
  const bool from_beam = n_hyps_host == nullptr;
  const int* d_tokens = nullptr;
  const double* d_scores = nullptr;
  if (from_beam) {
    // the last wn_ctc_prefix_beam_search of this handle: tokens and scores are still in its
    // device block, counts and lengths in the pinned copy of it
    WN_CHECK(m->pb_valid && m->pb_B == B,
             "wn_rescore: no prefix beam result of the current batch on this handle (pass the "
             "n-best explicitly, or call wn_ctc_prefix_beam_search first)");
    WN_CHECK(!hyp_lens_host && !hyp_tokens_host && !ctc_scores_host,
             "wn_rescore: n_hyps == NULL takes the whole n-best from the handle");
    WN_CHECK(beam == m->pb_beam && max_len == m->pb_max_len,
             "wn_rescore: beam / max_len differ from the prefix beam search's");
    n_hyps_host = reinterpret_cast<const int*>(m->pb_host.p + m->pb_o_nh);
    hyp_lens_host = reinterpret_cast<const int*>(m->pb_host.p + m->pb_o_len);
    d_tokens = reinterpret_cast<const int*>(m->pb_out.as<char>() + m->pb_o_tok);
    d_scores = reinterpret_cast<const double*>(m->pb_out.as<char>() + m->pb_o_sc);
  } else {
    WN_CHECK(hyp_lens_host && hyp_tokens_host && ctc_scores_host, "wn_rescore: null n-best");
  }
  WN_CHECK(beam >= 1 && beam <= 64 && max_len >= 1, "wn_rescore: beam must be 1..64");
  const bool use_r2l = reverse_weight > 0.0 && !m->right.layers.empty();
  // ---- ragged hypothesis batch: one sequence per (utterance, hypothesis) ----------------
  std::vector<int> seq_src, seq_first(B + 1), qoff, qlen, kvoff, kvlen;
  int R = 0, max_q = 0;
  for (int b = 0; b < B; ++b) {
    seq_first[b] = (int)seq_src.size();
    WN_CHECK(n_hyps_host[b] >= 0 && n_hyps_host[b] <= beam, "wn_rescore: n_hyps");
    if (n_hyps_host[b] > 0)
      WN_CHECK(m->len[b] > 0, "wn_rescore: utterance without encoder frames");
    for (int i = 0; i < n_hyps_host[b]; ++i) {
      const int L = hyp_lens_host[b * beam + i];
      WN_CHECK(L >= 0 && L <= max_len, "wn_rescore: hypothesis length");
      WN_CHECK(L + 1 <= c.max_pos, "wn_rescore: hypothesis longer than the positional table");
      if (!from_beam) {
        const int32_t* h = hyp_tokens_host + ((int64_t)b * beam + i) * max_len;
        for (int j = 0; j < L; ++j)
          WN_CHECK(h[j] >= 0 && h[j] < V, "wn_rescore: token id out of range");
      }
      seq_src.push_back(b * beam + i);
      qoff.push_back(R); qlen.push_back(L + 1);
      kvoff.push_back(m->off[b]); kvlen.push_back(m->len[b]);
      R += L + 1;
      max_q = std::max(max_q, L + 1);
    }
  }
  const int n_seq = (int)seq_src.size();
  seq_first[B] = n_seq;
  // cross-attention groups: all hypothesis rows of an utterance against its encoder frames
  std::vector<int> gq_off, gq_len, gkv_off, gkv_len;
  int g_max_q = 0;
  for (int b = 0; b < B; ++b) {
    const int q0 = seq_first[b], q1 = seq_first[b + 1];
    if (q1 <= q0) continue;
    const int rows = qoff[q1 - 1] + qlen[q1 - 1] - qoff[q0];
    gq_off.push_back(qoff[q0]); gq_len.push_back(rows);
    gkv_off.push_back(m->off[b]); gkv_len.push_back(m->len[b]);
    g_max_q = std::max(g_max_q, rows);
  }
  const size_t nb = (size_t)B * beam;
  WN_TRY(m->stage.begin((size_t)(5 * n_seq + 5 * B + 64) * sizeof(int) + 8192 +
                        (from_beam ? 0 : nb * max_len * sizeof(int) + nb * sizeof(double) + 256)));
  if (!from_beam) {
    // tokens | scores in one device block, the prefix beam search's own row pitch
    const size_t tok_bytes = (nb * max_len * sizeof(int) + 7) / 8 * 8;
    WN_TRY(m->r_hyp.ensure(tok_bytes + nb * sizeof(double)));
    WN_TRY(m->stage.put_at(m->r_hyp.p, hyp_tokens_host, nb * max_len * sizeof(int), s));
    WN_TRY(m->stage.put_at(m->r_hyp.as<char>() + tok_bytes, ctc_scores_host,
                           nb * sizeof(double), s));
    d_tokens = m->r_hyp.as<int>();
    d_scores = reinterpret_cast<const double*>(m->r_hyp.as<char>() + tok_bytes);
  }
  WN_TRY(upload_desc(m, m->r_seqsrc, seq_src, s));
  WN_TRY(upload_desc(m, m->r_seqfirst, seq_first, s));
  WN_TRY(upload_desc(m, m->r_qoff, qoff, s));
  WN_TRY(upload_desc(m, m->r_qlen, qlen, s));
  WN_TRY(upload_desc(m, m->r_kvoff, kvoff, s));
  WN_TRY(upload_desc(m, m->r_kvlen, kvlen, s));
  WN_TRY(upload_desc(m, m->r_gqoff, gq_off, s));
  WN_TRY(upload_desc(m, m->r_gqlen, gq_len, s));
  WN_TRY(upload_desc(m, m->r_gkvoff, gkv_off, s));
  WN_TRY(upload_desc(m, m->r_gkvlen, gkv_len, s));
  WN_TRY(m->stage.end(s));
  CrossGroups cgrp;
  cgrp.q_off = m->r_gqoff.as<int>(); cgrp.q_len = m->r_gqlen.as<int>();
  cgrp.kv_off = m->r_gkvoff.as<int>(); cgrp.kv_len = m->r_gkvlen.as<int>();
  cgrp.n_seq = (int)gq_off.size(); cgrp.max_q = g_max_q;
  const CrossGroups* cg = tune().rescore_groups != 0 && !gq_off.empty() ? &cgrp : nullptr;
  // results, one block: tok_conf | conf | best_score | best_idx | all_scores
  const size_t o_tc = 0, o_cf = o_tc + (size_t)B * max_len * sizeof(double),
               o_bs = o_cf + (size_t)B * sizeof(double), o_bi = o_bs + (size_t)B * sizeof(float),
               o_as = o_bi + (size_t)B * sizeof(int), o_end = o_as + nb * sizeof(float);
  WN_TRY(m->r_res.ensure(o_end));
  WN_TRY(m->r_host.ensure(o_end));
  WN_HIP(hipMemsetAsync(m->r_res.p, 0, o_end, s));
  const int Rp = std::max(R, 1);
  for (DevBuf* bf : {&m->r_tok, &m->r_rtok, &m->r_pos, &m->r_tgt, &m->r_rtgt})
    WN_TRY(bf->ensure((size_t)Rp * sizeof(int)));
  WN_TRY(m->r_out.ensure((size_t)2 * Rp * sizeof(float)));
  float* o_l = m->r_out.as<float>();
  float* o_r = o_l + Rp;
  if (n_seq > 0) {
    hipLaunchKernelGGL(rescore_rows_kernel, dim3(n_seq), dim3(64), 0, s, m->r_seqsrc.as<int>(),
                       m->r_qoff.as<int>(), m->r_qlen.as<int>(), d_tokens, max_len, c.sos, c.eos,
                       V, m->r_tok.as<int>(), m->r_rtok.as<int>(), m->r_pos.as<int>(),
                       m->r_tgt.as<int>(), m->r_rtgt.as<int>());
    WN_HIP(hipGetLastError());
    WN_TRY(m->r_x.ensure((size_t)R * d * sizeof(float)));
    WN_TRY(m->r_t1.ensure((size_t)R * d * sizeof(float)));
    WN_TRY(m->r_t2.ensure((size_t)R * d * sizeof(float)));
    WN_TRY(m->r_qkv.ensure((size_t)R * 3 * d * sizeof(float)));
    WN_TRY(m->r_h.ensure((size_t)R * c.dec_ffn_dim * sizeof(float)));
    WN_TRY(m->r_mem.ensure((size_t)m->rows * 2 * d * sizeof(float)));
    WN_TRY(m->r_logits.ensure((size_t)R * ((V + 3) / 4 * 4) * sizeof(float)));
    // cross-attention K | V projected while the prefix beam search ran (wn_rescore_prefetch)?
    const float* kv_l = nullptr;
    const float* kv_r = nullptr;
    if (m->kv_ready) {
      // an outstanding prefetch is ALWAYS ordered in front of this pass and consumed here:
      // usable or not (e.g. prefetched without the right-to-left decoder, rescored with it),
      // its GEMMs on the side stream read m->enc and write r_kv_all while the decoder pass
      // below would run beside them (round-4 advice)
      WN_HIP(hipStreamWaitEvent(s, m->side.e1, 0));
      if (m->kv_rows == m->rows && m->kv_nl == (int)m->left.layers.size() &&
          (!use_r2l || m->kv_nr == (int)m->right.layers.size())) {
        kv_l = m->r_kv_all.as<float>();
        kv_r = kv_l + (size_t)m->kv_nl * m->rows * 2 * d;
      } else {
        m->kv_ready = false;
      }
    }
    WN_TRY(run_decoder(m, m->left, R, n_seq, max_q, m->r_tok.as<int>(), m->r_tgt.as<int>(), o_l,
                       s, kv_l, cg));
    if (use_r2l)
      WN_TRY(run_decoder(m, m->right, R, n_seq, max_q, m->r_rtok.as<int>(),
                         m->r_rtgt.as<int>(), o_r, s, kv_r, cg));
  }
  char* rb = m->r_res.as<char>();
  RescoreArgs a;
  a.lp_l = o_l; a.lp_r = o_r;
  a.seq_first = m->r_seqfirst.as<int>(); a.seq_src = m->r_seqsrc.as<int>();
  a.qoff = m->r_qoff.as<int>(); a.qlen = m->r_qlen.as<int>();
  a.ctc_scores = d_scores; a.beam = beam; a.max_len = max_len; a.use_r2l = use_r2l ? 1 : 0;
  a.ctc_weight = ctc_weight;
  a.w_l = (float)(1.0 - reverse_weight); a.w_r = (float)reverse_weight;
  a.best_idx = reinterpret_cast<int*>(rb + o_bi);
  a.best_score = reinterpret_cast<float*>(rb + o_bs);
  a.conf = reinterpret_cast<double*>(rb + o_cf);
  a.all_scores = reinterpret_cast<float*>(rb + o_as);
  a.tok_conf = reinterpret_cast<double*>(rb + o_tc);
  hipLaunchKernelGGL(rescore_reduce_kernel, dim3(B), dim3(64), 0, s, a);
  WN_HIP(hipGetLastError());
  WN_HIP(hipMemcpyAsync(m->r_host.p, rb, o_end, hipMemcpyDeviceToHost, s));
  WN_HIP(stream_wait(s));
  const char* hb = m->r_host.p;
  memcpy(best_idx_host, hb + o_bi, (size_t)B * sizeof(int));
  memcpy(best_score_host, hb + o_bs, (size_t)B * sizeof(float));
  if (confidence_host) memcpy(confidence_host, hb + o_cf, (size_t)B * sizeof(double));
  if (tok_conf_host) memcpy(tok_conf_host, hb + o_tc, (size_t)B * max_len * sizeof(double));
  if (all_scores_host) memcpy(all_scores_host, hb + o_as, nb * sizeof(float));
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
  if (tune().x6_af32 != 0 && (int64_t)M * K * 4 < ((int64_t)1 << 31)) {
    a.A = A; a.lda = K; a.a_bytes = (int64_t)M * K * 4;      // split in registers
  } else {
    WN_TRY(a3.ensure(x6_bytes(M, K)));
    WN_TRY(x6_split(A, M, K, K, a3.as<char>(), s));
    a.A3 = a3.as<char>();
  }
  a.B3 = w3.as<char>(); a.M = M; a.N = N; a.K = K; a.bm = bm;
  if (bm == 120) { a.bm = 128; a.nw = 8; }   // micro-benchmark: the 8-wave 128-row tile
  if (bm == 129) { a.bm = 128; a.nw = 4; }   // ... the 4-wave form
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
  if (tune().ffn_x6f != 0 && tune().x6_af32 == 0 && ffn_x6f_supported(M, D, F, act)) {
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
    if (tune().ffn_ximg == 2) {     // tests / tools: X handed over as its plane image
      WN_TRY(x3.ensure(x6_bytes(M, D)));
      WN_TRY(x6_split(X, M, D, D, x3.as<char>(), s));
      a.X3 = x3.as<char>(); a.X = nullptr;
    }
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
  const bool af32 = tune().x6_af32 != 0 && (int64_t)M * F * 4 < ((int64_t)1 << 31);
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

int wn_op_gemm_x6r512(const float* A, const float* W, const float* bias, float* x_inout,
                      const float* ln_w, const float* ln_b, float* y, const float* W2,
                      const float* bias2, float* C, int32_t M, int32_t N, int32_t epi, float alpha,
                      float eps, int32_t reps, void* stream) {
  WN_CHECK(A && W && M > 0 && gemm_x6r512_supported(M, N, epi), "gemm_x6r512: shape");
  WN_CHECK(epi != 3 || (W2 && C), "gemm_x6r512: the chained epilogue needs W2 and C");
  hipStream_t s = (hipStream_t)stream;
  static thread_local DevBuf w3, w3b;
  WN_TRY(w3.ensure(x6_bytes(N, 512)));
  WN_TRY(x6_split(W, N, 512, 512, w3.as<char>(), s));
  X6RArgs a;
  a.A = A; a.lda = 512; a.K = 512; a.W3 = w3.as<char>(); a.bias = bias; a.M = M; a.N = N;
  a.epi = epi; a.C = C; a.ldc = epi == 3 ? 512 : N; a.resid = x_inout; a.ldr = N; a.alpha = alpha;
  a.x_out = x_inout; a.ldx = N; a.ln_w = ln_w; a.ln_b = ln_b; a.eps = eps; a.y = y; a.ldy = N;
  if (epi == 3) {
    WN_TRY(w3b.ensure(x6_bytes(1024, 512)));
    WN_TRY(x6_split(W2, 1024, 512, 512, w3b.as<char>(), s));
    a.W3b = w3b.as<char>(); a.bias2 = bias2;
  }
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

