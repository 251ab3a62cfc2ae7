// Tuning knobs of the library: A/B switches between kernel forms, measurement probes and
// test hooks.  The defaults are the shipped configuration.
//
// Every knob is read through tune(): inside a C-ABI call on a handle that is the handle's
// EFFECTIVE set -- its own override where wn_model_tune_set gave one, else the process default
// that wn_tune_set writes -- installed for the calling thread by WN_ENTER (model_state.h).
// Two handles on two host threads therefore never see each other's overrides.  The handle-less
// test operators (wn_op_*) and wn_model_create read the process defaults.
//
// Ablation values (wrong results by design) are refused by both setters unless the library was
// built with WN_ABLATION=1 (build.py).
#pragma once
#include <stdint.h>

#include <string>

namespace wn {

// X(name, default)
#define WN_TUNE_KEYS(X)                                                                         \
  /* bf16 GEMM tile rule (gemm_bf16{,s}.hip): 0 = the shape rule, 1 = 128-row tiles only,     \
     8 = force the pipelined 256 x 256 kernel (gemm_bf16p.hip); tests */                        \
  X(gemm_tile_bf16, 0)                                                                          \
  /* bf16 mode: 1 = the GEMM-only tensors (LayerNorm output, FFN hidden, attention context)    \
     are stored as bf16; 0 = every tensor stays fp32 and the GEMMs convert on the fly.         \
     Measured r01h: config 5 73.0 -> 58.5 ms, identical arithmetic */                          \
  X(bf16_store, 1)                                                                              \
  /* pipelined bf16 / MXFP8 kernel: 4 = clock stamps (tools/lp_clocks.py) */                    \
  X(lp_probe, 0)                                                                                \
  /* fp32 attention: waves split over the key range: 0 auto, 1 off, 2 on */                     \
  X(attn_split, 0)                                                                              \
  /* 1 = the bf16 mode uses the bf16 attention kernel (attention_bf16.hip) */                   \
  X(attn_bf16, 1)                                                                               \
  /* bf16 attention: waves (32-query groups) per block: 0 auto, else 2 / 4 / 8 */               \
  X(attn_bf16_nw, 0)                                                                            \
  /* bf16 attention: deferred-rescale threshold x 10 in log2 units (0 = rescale whenever a     \
     maximum moves) */                                                                          \
  X(attn_bf16_defer, 80)                                                                        \
  /* bf16 Q | K | V self attention: 0 = register-staged kernel (A/B, tests), else K / V rows   \
     by LDS-DMA + transpose reads */                                                            \
  X(attn_bf16_dma, 1)                                                                           \
  /* bf16-storage form, encoders without the rel-pos term: 1 = the QKV GEMM writes bf16 and    \
     the attention kernel reads it; 0 keeps fp32 Q / K / V (A/B, tests) */                      \
  X(qkv_bf16, 1)                                                                                \
  /* fp8 mode: smallest number of 256 x 256 tiles of an FFN GEMM pair for which the MXFP8      \
     kernels are used (below it the bf16 kernels fill the chip better); tests set 0 */          \
  X(fp8_min_tiles, 192)                                                                         \
  /* fp32 v_mfma_f32 fused FFN (ffn_fused.hip): 0 = the two-GEMM path, 2 = force (tests) */     \
  X(ffn_fused, 1)                                                                               \
  /* 1 = fp32 GEMMs run as six bf16 plane products (gemm_x6.hip); 0 = the v_mfma_f32 kernels   \
     (A/B, tests), 2 = force */                                                                 \
  X(gemm_x6, 1)                                                                                 \
  /* 0 = linear() never routes to the six-product GEMM */                                       \
  X(x6_linear, 1)                                                                               \
  /* 0 = activations reach gemm_x6 as plane images; 1 = as fp32 rows split in registers.       \
     Measured (r02ag): the split costs more than the plane bytes it saves */                    \
  X(x6_af32, 0)                                                                                 \
  /* subsampling only: 1 = conv1 writes fp32 pixels (4 B / element instead of the 6-B plane     \
     image) and conv2 splits them while it stages its A operand (A/B; r14k: slower) */          \
  X(x6_conv_af32, 0)                                                                            \
  /* wn_model_set_encode_gate: where wn_encode waits for the event: 0 = behind its descriptor   \
     uploads, in front of conv1; 1 = behind CMVN + conv1 (that kernel then runs beside the       \
     previous decode's layers: +0.7 % with the 256-row conv2, -0.6 % with the 128-row one) */   \
  X(enc_gate_pos, 0)                                                                            \
  /* subsampling conv2 (six-product implicit GEMM): 128 = one launch of 128-row tiles on four   \
     waves, two blocks per CU (loses least to the prefix beam search it shares the chip with     \
     when decodes are in flight); 0 = 256-row tiles on eight waves + the last round as K         \
     slices (3 % faster alone; A/B) */                                                          \
  X(x6_conv_bm, 128)                                                                            \
  /* gemm_x6: 4 = clock stamps; 8 = wn_profile_gemm_clocks returns the row-block kernel's      \
     phase stamps (gemm_x6r.hip); WN_ABLATION builds: 1 no MFMAs, 2 no DMA */                  \
  X(x6_probe, 0)                                                                                \
  /* fused six-product FFN (ffn_x6f.hip, d_model 256): 0 = the two six-product GEMMs (A/B,     \
     tests), 3 = round 4's rule (only batches that fill half the CUs; A/B) */                   \
  X(ffn_x6f, 1)                                                                                 \
  /* fused FFN input: 1 = the producers of LN(x) (pointwise_conv2 row-block GEMM, the partial   \
     reduction in front of the next layer) write its X3 plane image and the kernel loads         \
     fragments; 0 = fp32 rows split by every slice block (A/B, tests: bit-identical) */          \
  X(ffn_ximg, 1)                                                                                \
  /* WN_ABLATION builds: 4..6 = the older DMA stages of 24 records */                           \
  X(ffn_x6f_ring, 3)                                                                            \
  /* 25088 = the clock-stamp form; WN_ABLATION builds: the other VAR variants */                \
  X(ffn_x6f_var, 0)                                                                             \
  /* row-block six-product GEMMs (gemm_x6r{,512}.hip): 0 = the v_mfma_f32 row-LN GEMM / tile   \
     GEMMs (A/B, tests) */                                                                      \
  X(x6r, 1)                                                                                     \
  /* 0 = out-projection + LayerNorm and pointwise_conv1 + GLU as two launches */                \
  X(x6r_chain, 1)                                                                               \
  /* 0 = dwconv_ln_silu stays its own launch in front of pointwise_conv2 (A/B, tests) */        \
  X(x6r_dwc, 1)                                                                                 \
  /* 0 = ffn_reduce_ln stays its own launch in front of the QKV projection (A/B, tests) */      \
  X(x6r_pro, 1)                                                                                 \
  /* K = 512 row-block kernels: 0 auto, 32 / 64 force the block height (A/B, tests) */          \
  X(x6r512_rows, 0)                                                                             \
  /* 0 = GEMM + LayerNorm launches instead of the v_mfma_f32 row-LN GEMM (A/B) */               \
  X(gemm_rowln, 1)                                                                              \
  /* depthwise convolution: 1 = four rows per wave; 0 = one row per wave (A/B, tests) */        \
  X(dwconv_tiled, 1)                                                                            \
  /* rel-pos self attention of the fp32 mode over full-context batches: 1 = six bf16 plane      \
     products (attention_x6.hip: pack pass + kernel), 0 = the v_mfma_f32 kernel (A/B, tests);   \
     2 = also under chunk masks (measured slightly slower there) */                             \
  X(attn_x6, 1)                                                                                 \
  /* fp32 attention kernels: 1 = XCD-aware block order (the query blocks of a (sequence, head)  \
     share one XCD's L2), 0 = the plain 3-D grid (A/B, tests: bit-identical) */                 \
  X(attn_xcd, 1)                                                                                \
  /* six-product attention: 1 = key-tile images aligned to the global 32-row blocks of the      \
     packed K / V matrix (the row blocks of the QKV projection); 2 = ... and written by that     \
     projection's epilogue, no pack pass; 0 = aligned to each sequence's first key (A/B) */      \
  X(attn_x6_galign, 2)                                                                          \
  /* six-product attention of the encoder: 1 = its blocks are dispatched from a list -- two     \
     live query groups first, the light last blocks of odd sequences behind them -- 0 = the      \
     (query block, head, sequence) grid (A/B, tests: bit-identical) */                           \
  X(attn_x6_order, 1)                                                                           \
  /* WN_ABLATION builds: attention_x6_kernel without parts of itself (attention_x6.hip ABL) */  \
  X(attn_x6_var, 0)                                                                             \
  /* rel-pos attention: 0 = two contractions per score, 2 = the fold as a separate pass */      \
  X(attn_fold, 1)                                                                               \
  /* 0 = cross attention of the rescoring decoder per hypothesis (A/B, tests) */                \
  X(rescore_groups, 1)                                                                          \
  /* 0 = wn_rescore_prefetch does nothing (A/B) */                                              \
  X(rescore_prefetch, 1)                                                                        \
  /* tests: 2-bit prefix hash in the prefix beam search (exercises the exact sequence test) */  \
  X(beam_weak_hash, 0)                                                                          \
  /* prefix beam search: 0 = the node pool stays in global scratch even where it fits LDS      \
     (A/B, tests) */                                                                            \
  X(beam_lds_pool, 1)                                                                           \
  /* result waits of the searches / rescoring: 1 = poll the stream, 0 = blocking wait (A/B) */  \
  X(sync_spin, 1)                                                                               \
  /* CTC log-softmax: 0 = always the block-per-row kernel (A/B, tests) */                       \
  X(ctc_wave, 1)

struct Tune {
#define X(name, dflt) int name = dflt;
  WN_TUNE_KEYS(X)
#undef X
};

constexpr int32_t TUNE_INHERIT = INT32_MIN;   // per-handle override: "follow the process default"

extern Tune g_tune_default;                // wn_tune_set
extern thread_local const Tune* t_tune;    // effective set of the handle in this call, else null
inline const Tune& tune() { return t_tune ? *t_tune : g_tune_default; }

// key -> field; nullptr for an unknown key
int* tune_field(Tune& t, const std::string& key);
// checks the value (ablation gating); 0, or -1 with set_error
int tune_check(const std::string& key, int32_t value, const char* who);
// eff = ovr where set, else the process default
void tune_resolve(const Tune& ovr, Tune* eff);
Tune tune_all_inherit();

}  // namespace wn
