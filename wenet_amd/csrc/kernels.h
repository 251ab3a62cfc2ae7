// Launch wrappers of the non-GEMM kernels of libwenet_amd.
#pragma once
#include "common.h"

namespace wn {

// y[r,:] = LayerNorm(x[r,:]) * w + b, one wave per row; D in {64..1024, %64}.
// y_bf16: y is a bf16 matrix (ldy in bf16 elements) -- bf16-storage mode.
int layernorm(const float* x, int ldx, const float* w, const float* b, float* y,
              int ldy, int M, int D, float eps, hipStream_t s, bool y_bf16 = false);

// y = LayerNorm(x) written as MXFP8: q (M, D) e4m3 bytes + block scales
// [D/128][pitch] dwords (csrc/mxfp8.h); D % 256 == 0.
int layernorm_mx(const float* x, int ldx, const float* w, const float* b, void* q,
                 unsigned* scale, int pitch, int M, int D, float eps, hipStream_t s);

// y1 = LN(x; w1,b1), y2 = LN(y1; w2,b2) in one pass over contiguous [M][D] rows
// (y1 may alias x).
int layernorm2(const float* x, const float* w1, const float* b1, const float* w2,
               const float* b2, float* y1, float* y2, int M, int D, float eps,
               hipStream_t s, bool y2_bf16 = false);

// GlobalCMVN + Conv2d(1->C,3,stride 2) + ReLU over the padded (B,T,F) frame
// tensor, written packed channels-last: out[(t1_off[b]+t1)*F1 + f1][c].
struct Conv1Args {
  const float* feats;    // (B, T, F) padded
  const float* mean;     // [F] or null (no CMVN)
  const float* istd;     // [F]
  const float* w;        // [9][C]  (tap-major, reordered from (C,1,3,3))
  const float* bias;     // [C]
  float* out;            // [sum T1_b][F1][C]
  // instead of `out`: plane image (gemm_x6.hip) with one row per pixel, `tiles` 32-pixel
  // tiles; pixel = frame * F1 + pos, even f1 first (pos = f1 / 2), odd ones behind them
  char* out3 = nullptr;
  int tiles = 0;
  const int* t1_off;     // [B] packed row offset (in T1 frames)
  const int* t1_len;     // [B] number of T1 frames to produce
  int B, T, F, F1, C, max_t1;
};
int cmvn_conv1_relu(const Conv1Args& a, hipStream_t s);

// Depthwise conv over time + LayerNorm(channels) + SiLU (the middle of
// ConvolutionModule.forward).  x is the GLU output [rows][D].
struct DwConvArgs {
  const float* x; int ldx;
  const float* wt;      // [K][D] tap-major depthwise weights
  const float* bias;    // [D]
  const float* cpad;    // [D] value of a padded (masked / left-pad) frame
  const float* ln_w; const float* ln_b;
  int norm_mode = 0;    // 0: LayerNorm over channels (ln_w, ln_b); 1: per-channel
                        // affine y = x * ln_w + ln_b (eval-mode BatchNorm1d)
  float* y; int ldy;
  const int* row_utt;   // [M] utterance of each row (-1: skip)
  const int* off;       // [B] first row of utterance
  const int* len;       // [B] valid frames
  int M, D, K, causal, t_max;  // t_max: padded length T' of the batch
  float eps;
};
int dwconv_ln_silu(const DwConvArgs& a, hipStream_t s);

// Multi-head attention with online softmax (d_k = 64).  Optional relative
// position term: score = ((q+u).k + (q+v).p) * scale.
struct AttnArgs {
  const float* Q; const float* K; const float* V;
  int ldq, ldk, ldv;
  const float* P = nullptr; int ldp = 0;      // [T][h*64] projected pos table
  const int* p_off = nullptr;  // [n_seq] or null: key j of sequence s uses row p_off[s] + j
  const float* bias_u = nullptr; const float* bias_v = nullptr;  // [h][64]
  // folded rel-pos form (relpos_fold): K already holds k + p, kbias [key rows][n_heads] is
  // added to the score before the scale; P / bias_u / bias_v stay null
  const float* kbias = nullptr;
  // fold = true with P / bias_u / bias_v set: the same folding done INSIDE the fp32 attention
  // kernel while the keys are staged (no separate pass, K untouched)
  bool fold = false;
  float* O; int ldo;
  const int* q_off; const int* q_len;   // [n_seq]
  const int* kv_off; const int* kv_len; // [n_seq]
  int n_seq, n_heads, max_q_len;
  bool o_bf16 = false; // bf16 kernel only: O is a bf16 matrix (ldo in elements)
  bool qkv_bf16 = false;  // bf16 kernel only, no rel-pos: Q / K / V are bf16 (ld* in elements)
  float defer_thr = 0.f;   // DMA-staged bf16 kernel: deferred-rescale threshold (log2 units, 0 = off)
  int mask_mode = 0;   // 0: keys < kv_len; 1: causal; 2: chunk window
  int chunk_size = 0, left_chunks = -1;
  float scale = 0.125f;
  // scratch for the six-product form (attention_x6.hip): the key-tile images of this launch;
  // x6_rows = rows of the K / V matrix (all sequences)
  void* x6_img = nullptr; size_t x6_img_bytes = 0; int x6_rows = 0;
  // set by the launchers (tune attn_xcd): > 0 = the grid is 1-D and block b stands for logical
  // block xcd_block_order(b): all query blocks of a (sequence, head) on ONE XCD, whose L2 then
  // fetches that pair's keys / values once instead of once per query block; the value is the
  // number of query blocks per (sequence, head)
  int xcd_nqb = 0;
  // six-product form, key-tile images aligned to the GLOBAL 32-row blocks of the packed K / V
  // matrix (tile t = rows 32 t .. 32 t + 31, whatever sequences they belong to; the kernel
  // masks the slots outside its own sequence) instead of to each sequence's first key: the
  // tiles are then exactly the row blocks of the QKV projection, whose epilogue can write
  // them (x6_img_ready: the image is already there, no pack pass).  row_utt [x6_rows]: the
  // sequence of every row (pack pass with x6_galign)
  int x6_galign = 0;
  // six-product form: the launch's blocks in dispatch order, (sequence << 16 | head << 8 |
  // query block of 64) -- full blocks first, the light last blocks of odd sequences behind
  // them (model.hip set_layout); null = the (query block, head, sequence) grid
  const int* blk_tab = nullptr; int n_blk = 0;
  bool x6_img_ready = false;
  const int* row_utt = nullptr;
};
int attention(const AttnArgs& a, hipStream_t s);
// rel-pos self attention as six bf16 plane products (attention_x6.hip): pack pass + kernel
size_t attention_x6_image_bytes(int rows, int n_seq, int n_heads);
bool attention_x6_supported(const AttnArgs& a);
int attention_x6(const AttnArgs& a, hipStream_t s);
int relpos_fold(float* K, int ldk, const float* P, int ldp, const float* bias_u,
                const float* bias_v, const int* row_utt, const int* off, const int* p_off,
                float* kbias, int n_heads, int M, int D, hipStream_t s);
// The same with bf16 MFMA operands (attention_bf16.hip); attention() routes here
// when the calling thread's precision is PREC_BF16 (and tune().attn_bf16 != 0).
int attention_bf16(const AttnArgs& a, hipStream_t s);

// Fused feed-forward module, fp32 (ffn_fused.hip): P[s] (S, M, D) = partial
// act(X W1^T + b1) W2^T over hidden slice s; ffn_reduce_ln then forms
// x += alpha (sum_s P[s] + b2) and the following LayerNorm(s).
struct FfnArgs {
  const float* X;     // [M][D] = LayerNorm(x)
  const float* W1;    // [F][D]
  const float* b1;    // [F]
  const float* W2;    // [D][F]
  float* P;           // [S][M][D]
  int M, D, F, S, act;
};
int ffn_fused_split(int M, int D, int F);
bool ffn_fused_supported(int M, int D, int F, int act);
int ffn_fused(const FfnArgs& a, hipStream_t s);
// mode 0: x <- x_new, y <- LN(x_new; w, b); mode 1: x <- LN(x_new; w, b),
// y <- LN(x; w2, b2); mode 2: x <- LN(x_new; w, b)
int ffn_reduce_ln(float* x, const float* P, int S, const float* b2, float alpha,
                  const float* w, const float* b, const float* w2, const float* bb2, float* y,
                  int M, int D, float eps, int mode, hipStream_t s);
// mode 1 with y leaving as the X3 plane image (M x 256) instead of fp32 rows: the same
// arithmetic per row, eight rows per block, the planes turned through LDS into whole
// 128-byte pieces of the image
int ffn_reduce_ln_img(float* x, const float* P, int S, const float* b2, float alpha,
                      const float* w, const float* b, const float* w2, const float* bb2,
                      void* y3, int M, int D, float eps, hipStream_t s);

// fp32 GEMM as six bf16 plane products (gemm_x6.hip).  Operands are "X3" images:
// x6_bytes(R, K) bytes for an R x K matrix, made by x6_split or by a GEMM's EPI 2.
struct X6Args {
  const void* A3 = nullptr;   // image of A (M x K)
  // instead of A3: A as plain row-major fp32 (lda floats per row, a_bytes = extent of the
  // buffer), split into planes in registers; with a_pix: the channels-last fp32 tensor
  const float* A = nullptr; int lda = 0; int64_t a_bytes = 0;
  const void* B3 = nullptr;   // image of W (N x K)
  int M = 0, N = 0, K = 0;
  int row0 = 0;               // first row of C this launch computes (multiple of 256)
  int ksplit = 1;             // K slices (epi 1 only)
  int bm = 0;                 // block rows 128 / 256, 0 = auto
  int nw = 0;                 // 128-row tiles: four waves, two blocks per CU; 8 forces the 8-wave form
  int epi = 0;                // 0: C = resid + alpha act(acc + bias); 1: P[slice][M][N] = acc;
                              // 2: C3 = X3 image of act(acc + bias)
  const float* bias = nullptr;
  const float* resid = nullptr; int ldr = 0;
  float alpha = 1.0f; int act = 0;
  float* C = nullptr; int ldc = 0;
  void* C3 = nullptr;
  // gathered A (implicit GEMM of a convolution): A3 is the image of the channels-last input
  // with one row per pixel (a_tiles 32-pixel tiles, conv_kbc = C / 16 k blocks per tap);
  // GEMM row r reads pixel a_pix[r] + tap_delta[tap]
  const int* a_pix = nullptr;
  int a_tiles = 0, conv_kbc = 0;
  int conv_taps = 0;          // > 0: K order (channel block, tap) instead of (tap, channel block)
  int tap_delta[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int probe = 0;              // ablation bits (tune().x6_probe)
  // gathered A (conv2): scratch for the K-slice partials of the last, partial round of tiles
  // (part_bytes >= slices x rows x N x 4); null: the remainder runs as 128-row tiles
  float* part = nullptr; size_t part_bytes = 0;
};
int gemm_x6_clocks(unsigned long long* out);   // probe & 4 stamps [8 waves][8]
size_t x6_bytes(int R, int K);
int x6_split(const float* src, int R, int K, int ld, void* dst, hipStream_t s);
int gemm_x6_bm(int M, int N, int ksplit);
int gemm_x6(const X6Args& a, hipStream_t s);
// Fused six-product feed-forward module (ffn_x6f.hip): hidden tensor in registers, d_model 256.
struct FfnX6Args {
  const float* X = nullptr;    // X = LayerNorm(x): [M][ldx] fp32 (split into planes in registers)
  int ldx = 0;
  // instead of X (round 5): the X3 plane image of LN(x) (M x 256, records [k block][row tile]
  // [plane]) written by the producer of LN(x) -- the epilogue of the pointwise_conv2 row-block
  // GEMM, ffn_reduce_ln_img -- so that the S hidden-slice blocks of a row tile load ready-made
  // operand fragments instead of each loading, turning and splitting the fp32 rows
  const void* X3 = nullptr;
  const void* W13 = nullptr;   // X3 image of W1: F x 256
  const void* W2p = nullptr;   // k-slot-permuted image of W2 (x6_split_perm): 256 x F
  const float* b1 = nullptr;   // [F]
  float* P = nullptr;          // [S][M][256] hidden-slice partials
  int M = 0, D = 0, F = 0, S = 0, act = 0;
};
int ffn_x6f_clocks(unsigned long long* out);   // VAR & 8192 stamps [4 waves][24]
int x6_split_perm(const float* src, int R, int K, int ld, void* dst, hipStream_t s);
int ffn_x6f_split(int M, int F);
bool ffn_x6f_supported(int M, int D, int F, int act);
int ffn_x6f(const FfnX6Args& a, hipStream_t s);


// x_out = resid + alpha (A W^T + bias), y = LayerNorm(x_out): GEMM with N = 256 whose
// block owns complete rows (gemm_rowln.hip)
struct RowLnArgs {
  const float* A; int lda;
  const float* W;         // [256][K]
  const float* bias;      // [256] or null
  const float* resid; int ldr;   // may alias x_out
  float alpha;
  float* x_out; int ldx;
  const float* ln_w; const float* ln_b; float eps;
  float* y; int ldy;      // may alias A (a block reads only the rows it writes)
  int M, N, K;
};
// Row-block six-product GEMM, K = 256, A rows in registers (gemm_x6r.hip)
struct X6RArgs {
  const float* A = nullptr; int lda = 0;     // [M][lda] fp32
  int K = 256;                               // 256 (gemm_x6r.hip) or 512 (gemm_x6r512.hip)
  const void* W3 = nullptr;                  // X3 image of W (N x K)
  const float* bias = nullptr;               // [N] or null
  int M = 0, N = 0;
  int epi = 0;          // 0: C = acc + bias; 1: x_out = resid + alpha (acc + bias), y = LN(x_out);
                        // 2: C = GLU(acc + bias), N / 2 columns (W rows permuted [32 a | 32 gate])
  float* C = nullptr; int ldc = 0;
  const float* resid = nullptr; int ldr = 0; float alpha = 1.0f;   // resid may alias x_out
  float* x_out = nullptr; int ldx = 0;
  const float* ln_w = nullptr; const float* ln_b = nullptr; float eps = 1e-5f;
  float* y = nullptr; int ldy = 0;           // may alias A (a block reads its rows first)
  // epi 1, K = 256: additionally / instead (y may be null then) the X3 plane image of y (M x N,
  // ceil(M / 32) row tiles) for a six-product consumer that wants fragments, not rows
  void* y3 = nullptr;
  // epi 3 = epi 1 chained with C = GLU(y W3b^T + bias2): W3b = X3 image of a 2 K x K weight
  // (rows [32 values | 32 gates] per 64), C [M][K]; y is stored only if set
  const void* W3b = nullptr; const float* bias2 = nullptr;
  // prologue fold (epi 0, N = 3 K -- the QKV projection behind a fused feed-forward module): the
  // A rows are NOT read from `A` but formed as ffn_reduce_ln (mode 0) forms them,
  //   x_new = pro_x + pro_alpha (sum_s pro_P[s] + pro_b2),  A = LayerNorm(x_new; ln_w, ln_b, eps),
  // x_new written back to pro_x ([M][K]); pro_P = [pro_S][M][K] slice partials
  const float* pro_P = nullptr; int pro_S = 0; const float* pro_b2 = nullptr;
  float pro_alpha = 0.f; float* pro_x = nullptr;
  // depthwise-conv prologue (epi 1, K = N = 256 -- pointwise_conv2 behind the middle of the
  // convolution module): the A rows are NOT read from `A` but formed as dwconv_ln_silu forms
  // them from dw.x (dw.y is ignored: the rows never reach HBM)
  int dw_on = 0;
  DwConvArgs dw;
  // epi 4 (K = 256, N = 768, with the prologue fold): the QKV projection of a layer whose self
  // attention runs as six plane products over GLOBAL-row-aligned key tiles (attention_x6.hip,
  // AttnArgs::x6_galign): W3 / bias hold the rows permuted per head ([Q_h | K_h | V_h] x 64,
  // h = 0..3 -- wave h of a block owns head h), Q goes to C (columns h 64 ..) as fp32 rows, and
  // the block's 32 rows ARE key tile blockIdx.x: K' = k + p planes, V^T planes and the per-key
  // scalars u.k + v.p of the four heads are written straight into the tile image (at_img),
  // the pack pass's arithmetic in its order -- bit-identical, one launch and 42 MB of traffic
  // per layer less.  at_P [T][at_ldp] = the layer's projected position table, at_u / at_v
  // [4][64], at_row_utt [M] / at_off [n_seq] / at_p_off (or null) give a row its position
  void* at_img = nullptr;
  const float* at_P = nullptr; int at_ldp = 0;
  const float* at_u = nullptr; const float* at_v = nullptr;
  const int* at_row_utt = nullptr; const int* at_off = nullptr; const int* at_p_off = nullptr;
};
int gemm_x6r_clocks(unsigned long long* out);   // phase stamps of the last x6r_kernel launch (gemm_x6r.hip)
bool gemm_x6r_supported(int M, int N, int K, int epi);
int gemm_x6r(const X6RArgs& a, hipStream_t s);          // dispatches on a.K
bool gemm_x6r512_supported(int M, int N, int epi);
int gemm_x6r512(const X6RArgs& a, hipStream_t s);
bool gemm_rowln_supported(int M, int N, int K);
int gemm_rowln(const RowLnArgs& a, hipStream_t s);

// CTC head tail: per row log-softmax statistics + top-k (descending, lower
// index first on ties) (+ optionally the full log-prob row).
struct CtcRowArgs {
  const float* logits; int ld;    // [M][V]
  int M, V, k;
  int blank; float blank_penalty;
  float* topk_val;   // [M][k] log-probs
  int* topk_idx;     // [M][k]
  float* logp;       // [M][ld_out] or null
  int ld_out;
};
int ctc_logsoftmax_topk(const CtcRowArgs& a, hipStream_t s);

// Greedy collapse (ctc_greedy_search): per utterance remove repeats + blanks.
// filter_blank_embedding (asr_model.py:153-180): rows whose CTC arg-max is not 0, compacted
int nonblank_map(const int* top1, int stride, const int* off, const int* len, int B, int* map,
                 int* n_keep, hipStream_t s);
int nonblank_gather(const float* src, const int* map, const int* off, const int* n_keep,
                    const int* noff, const int* nlen, const int* row_utt_new, float* dst, int D,
                    int rows_new, hipStream_t s);
int ctc_greedy_collapse(const int* top1, int top1_stride, const int* off,
                        const int* len, int B, int blank, int* out_tokens,
                        int out_stride, int* out_lens, hipStream_t s);

// CTC prefix beam search, one workgroup per utterance.
// Flattened ContextGraph (wenet/utils/context_graph.py): node 0 is the root;
// the trie edges live in an open-addressing hash table keyed by
// (state << 32 | token), linear probing, `mask` + 1 slots, empty = ~0.
struct CtxGraph {
  const int* fail = nullptr;              // [n_nodes]
  const double* node_score = nullptr;     // [n_nodes]
  const double* output_score = nullptr;   // [n_nodes]
  const double* token_score = nullptr;    // [n_nodes]
  const unsigned long long* keys = nullptr;
  const int* vals = nullptr;
  unsigned mask = 0;
};
constexpr unsigned long long CTX_EMPTY = ~0ull;
__host__ __device__ inline unsigned ctx_slot(unsigned long long key, unsigned mask) {
  unsigned long long z = key * 0x9E3779B97F4A7C15ull;
  z ^= z >> 29;
  return (unsigned)z & mask;
}

struct PrefixBeamArgs {
  const float* topk_val; const int* topk_idx; int k;  // [rows][k]
  const int* off; const int* len; int B;
  int beam, blank, max_len;
  // node pools (device scratch): see prefix_beam.hip
  int* pool; int64_t pool_stride;  // ints per utterance
  // outputs
  int* n_hyps;          // [B]
  int* hyp_lens;        // [B][beam]
  int* hyp_tlens;       // [B][beam] length of the times list
  int* hyp_tokens;      // [B][beam][max_len]
  int* hyp_times;       // [B][beam][max_len]
  double* hyp_scores;   // [B][beam]
  // optional phase timing of workgroup 0 (s_memtime cycles, summed over
  // frames): [0] eval, [1] rank, [2] select/write, [3] frames, [4] emit
  long long* dbg_cycles = nullptr;
  CtxGraph cg;  // keys == nullptr: no context biasing
  int weak_hash = 0;  // tests: 2-bit prefix hash, so that the exact sequence test decides
};
int64_t prefix_beam_pool_ints(int max_len, int beam);
// out[i] = log_add(a[i], b[i]) with the search's own fp64 routine (parity test)
int log_add_pairs(const double* a, const double* b, double* out, int n,
                  hipStream_t s);
int ctc_prefix_beam(const PrefixBeamArgs& a, hipStream_t s);

// Kaldi fbank (see fbank.hip).
struct FbankArgs {
  const float* pcm;          // all utterances back to back, float in [-1, 1]
  const int64_t* sample_off; // [B] first sample of each utterance (device)
  const int* n_frames;       // [B] (device)
  int B, max_frames, n_mel;
  const float* window;       // [400] povey
  const float* twiddle;      // [256][2] cos, -sin of 2*pi*k/512
  const int* mel_start; const int* mel_len; const int* mel_off;  // [n_mel]
  const float* mel_w;        // CSR weights
  float* feats;              // (B, max_frames, n_mel)
};
int fbank_kaldi(const FbankArgs& a, hipStream_t s);
// polyphase sinc resampler (see fbank.hip); taps [nnew][K]
int resample_sinc(const float* x, int64_t n_in, const float* taps, int K, int width,
                  int orig, int nnew, float* out, int64_t n_out, hipStream_t s);

// Whisper log-mel (see logmel.hip): K of the DFT GEMM (400 padded to 416), width
// of its output (201 cos + 201 sin rows), K of the mel GEMM (201 padded to 224).
constexpr int LOGMEL_K1 = 416;
constexpr int LOGMEL_NS = 402;
constexpr int LOGMEL_K2 = 224;
struct LogMelArgs {
  const float* pcm;            // waveforms back to back, float in [-1, 1]
  const int64_t* sample_off;   // [B + 1] (device)
  const int* row_utt;          // [rows] utterance of each packed frame
  const int* frame_off;        // [B] first packed frame of each utterance
  const float* window;         // [400] periodic hann
  float* frames;               // [rows][LOGMEL_K1]
};
int logmel_frames(const LogMelArgs& a, int rows, hipStream_t s);
int logmel_power(const float* spec, float* pw, int rows, hipStream_t s);
int logmel_finish(float* mel, int n_mels, const int* frame_off, const int* n_frames,
                  float* umax, int B, int max_frames, float* feats, hipStream_t s);

// `attention` decode mode, device side (attn_search.hip)
int attn_step_embed(const int* last_tok, int pos, const float* emb, const float* pe,
                    float scale, int d, int n, float* x, hipStream_t s);
int attn_self_step(const float* qkv, int d, int heads, int n, float* cache, int step,
                   const int* path, int max_len, float* out, hipStream_t s);
int attn_beam_init(int BN, int N, int max_len, int sos, float* score, int* end, int* tok,
                   int* path, int* last_tok, hipStream_t s);
int attn_beam_update(int B, int N, int step, int max_len, int eos, int V, const float* topv,
                     const int* topi, const float* score_in, const int* end_in,
                     const int* tok_in, const int* path_in, float* score_out, int* end_out,
                     int* tok_out, int* path_out, int* last_tok, int* n_done, hipStream_t s);
int attn_beam_finish(int B, int N, int len, int max_len, int eos, float length_penalty,
                     const float* score, const int* tok, int* out_tok, int* out_len,
                     hipStream_t s);

// rows scatter/gather helpers
// forward_chunk cache plumbing (see encoder_kernels.hip), one descriptor per streaming
// session of the call.  The cache tensors keep the reference's per-session layouts:
// att (n_layers, heads, t1, 128), cnn (n_layers, 1, d, lorder).
struct ChunkSess {
  const float* att_cache;   // null iff t1 == 0
  float* new_att;           // (n_layers, heads, nt, 128)
  const float* cnn_cache;   // null: zeros (first chunk)
  float* new_cnn;
  int t1, next_start, nt, kv_off;   // kv_off: first row of this session in the K|V buffer
};
int chunk_kv_assemble(const ChunkSess* sess, int n_sess, int layer, int max_tk,
                      const float* qkv, int R, int H, float* kv, hipStream_t s);
int chunk_conv_input(const ChunkSess* sess, int n_sess, int layer, const float* x, int R,
                     int d, int lorder, float* xext, hipStream_t s);
int copy_rows(const float* src, int lds, const int* src_rows, float* dst,
              int ldd, const int* dst_rows, int n_rows, int D, hipStream_t s);
int fill_zero(void* p, size_t bytes, hipStream_t s);

}  // namespace wn
