/* libwenet_amd -- C ABI of the MI355X-native Conformer-ASR inference path.
 *
 * The reference (wenet-e2e/wenet) has NO operator / FFI boundary on this path:
 * `ASRModel.decode()` (wenet/models/transformer/asr_model.py:267-343) calls
 * torch.nn modules and the pure-Python functions of
 * wenet/models/transformer/search.py.  This header is the boundary a
 * maintainer would bind instead (ctypes stub in INTEGRATION.md); every entry
 * point names the reference function(s) it replaces.  The only C-ABI precedent
 * in the reference is the streaming recogniser runtime/core/api/wenet_api.h:27-108
 * (opaque handle, int status); the same style is kept here, but batch-oriented
 * and tensor-in / tensor-out.
 *
 * Conventions
 *  - every function returns 0 on success, <0 on error (-1 bad argument, -2 HIP
 *    error, -3 missing / mis-shaped weight, -4 handle busy); wn_last_error() gives the thread-local message;
 *  - `*_dev` pointers are device (HBM) pointers owned by the caller (e.g. a
 *    torch-ROCm tensor's data_ptr()); `*_host` pointers are host memory;
 *  - `stream` is a hipStream_t (0 = default stream).  Launches are
 *    asynchronous; functions that fill host outputs synchronise `stream`
 *    before returning;
 *  - the library owns the weights and a workspace arena that grows on demand;
 *    one wn_model per process/GPU (one process per GPU for multi-GPU);
 *  - a handle is used by ONE host thread at a time (workspace, descriptor
 *    staging and current batch are per-handle state): a second thread that
 *    enters a busy handle gets -4 and must use its own wn_model_clone();
 *  - all arithmetic is fp32 (fp64 for the prefix-beam bookkeeping, like the
 *    reference's Python floats) unless wn_model_set_precision() opts a handle
 *    into bf16 operands; tokens / lengths are int32.
 */
#ifndef WENET_AMD_H_
#define WENET_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wn_model wn_model;

/* Parsed train.yaml (wenet/utils/init_model.py:100-181 reads the same keys). */
typedef struct {
  int32_t feat_dim;          /* input_dim, 80 */
  int32_t d_model;           /* encoder_conf.output_size */
  int32_t n_heads;           /* encoder_conf.attention_heads (d_model/n_heads must be 64) */
  int32_t ffn_dim;           /* encoder_conf.linear_units */
  int32_t n_layers;          /* encoder_conf.num_blocks */
  int32_t cnn_kernel;        /* encoder_conf.cnn_module_kernel */
  int32_t causal;            /* encoder_conf.causal */
  int32_t use_dynamic_chunk; /* encoder_conf.use_dynamic_chunk */
  int32_t static_chunk_size; /* encoder_conf.static_chunk_size */
  int32_t vocab;             /* output_dim */
  int32_t has_cmvn;
  int32_t dec_heads;         /* decoder_conf.attention_heads */
  int32_t dec_ffn_dim;       /* decoder_conf.linear_units */
  int32_t dec_layers;        /* decoder_conf.num_blocks (left-to-right) */
  int32_t dec_r_layers;      /* decoder_conf.r_num_blocks; 0 if not bitransformer */
  int32_t bidirectional;     /* decoder == 'bitransformer' */
  int32_t sos, eos;          /* asr_model.py:59-62 */
  int32_t max_pos;           /* positional table length (5000) */
  float norm_eps;            /* 1e-5 */
  /* encoder family (wenet/utils/init_model.py:52-97, class_utils.py:37-98) */
  int32_t encoder_type;      /* 0: conformer (rel_pos, conv2d, swish, macaron)
                                1: transformer (TransformerEncoderLayer,
                                   encoder_layer.py:28-127; Whisper encoder) */
  int32_t input_layer;       /* 0: conv2d (Conv2dSubsampling4)
                                1: conv1d2 (Conv1dSubsampling2, subsampling.py:117-171) */
  int32_t activation;        /* FFN activation: 0 swish/SiLU, 1 gelu (exact erf) */
  int32_t key_bias;          /* attention linear_k has a bias (Whisper: 0) */
  int32_t cnn_norm;          /* conv-module norm: 0 layer_norm, 1 batch_norm (eval:
                                running statistics; convolution.py:77-81) */
} wn_config;

/* One entry of the reference state_dict (fp32, host memory, C-contiguous). */
typedef struct {
  const char* name;
  const float* data;
  int64_t numel;
} wn_tensor;

const char* wn_last_error(void);
const char* wn_version(void);

/* Build a model from a reference state_dict; replaces
 * init_model + load_checkpoint + model.to(device)
 * (wenet/utils/init_model.py:184, wenet/utils/checkpoint.py:26-43,
 *  wenet/cli/model.py:109).  Weights are re-laid-out for the kernels. */
int wn_model_create(const wn_config* cfg, const wn_tensor* weights,
                    int32_t n_weights, int32_t device, wn_model** out);
void wn_model_destroy(wn_model* m);

/* A second handle on the same (immutable, shared) weights with its own
 * workspace and current batch: one handle per in-flight batch lets a host keep
 * several batches running on different streams (wenet_amd/pipeline.py).
 * Handles may be used from different host threads, one thread per handle. */
int wn_model_clone(const wn_model* src, wn_model** out);

/* Operand precision of the handle's contractions: the reference's
 * `recognize.py --dtype {fp32,bf16}` (wenet/bin/recognize.py:52-56,250-255: torch
 * autocast around model.decode).
 *   WN_PREC_F32  (default) fp32 operands, fp32 accumulation: the parity mode (identical
 *                greedy tokens, rescoring scores within 1e-3).  Large contractions run
 *                as six products of the operands' three exact bf16 planes on the bf16
 *                matrix cores (error against fp64 not above v_mfma_f32's:
 *                tests/test_gpu_x6.py); small ones, and everything under
 *                wn_tune_set("gemm_x6", 0), on v_mfma_f32_32x32x2_f32;
 *   WN_PREC_BF16 every Linear / pointwise-conv / subsampling-conv contraction
 *                rounds its two operands to bf16 (round to nearest even) and
 *                accumulates in fp32; the attention products likewise; LayerNorm,
 *                softmax, the depthwise conv and the searches stay fp32
 *                (autocast additionally rounds every result to bf16, so this
 *                mode is at least as precise as the reference's).  Tensors that
 *                only feed such a contraction may be KEPT as bf16 in HBM -- same
 *                values, the contraction rounds them first thing.
 *   WN_PREC_FP8  WN_PREC_BF16, and the two GEMMs of every encoder feed-forward module
 *                (w_1, w_2, positionwise_feed_forward.py:50-58) on OCP MXFP8 operands:
 *                e4m3 elements with one power-of-two (E8M0) scale per 32 consecutive
 *                k, for activations (quantised by the LayerNorm / by the w_1
 *                epilogue) and weights (quantised once here), multiplied on the
 *                block-scaled fp8 matrix cores with fp32 accumulation -- BASELINE.json
 *                configs[4] "MFMA fp8 FFN".  Shapes too small to fill the chip with
 *                256 x 256 tiles stay on the bf16 kernels.
 * Applies to later calls on this handle; clones inherit it.  The feature
 * frontends (wn_fbank, wn_log_mel, wn_resample) always run in fp32. */
#define WN_PREC_F32 0
#define WN_PREC_BF16 1
#define WN_PREC_FP8 2
int wn_model_set_precision(wn_model* m, int32_t precision);
int32_t wn_model_get_precision(const wn_model* m);

/* Number of utterances of the handle's CURRENT batch (what the last wn_encode /
 * wn_set_encoder_out / wn_set_ctc_probs installed; 0 before any, -1 for a null handle): the
 * per-utterance output arrays of the searches and of wn_rescore are sized by it.  No
 * reference counterpart -- the reference reads encoder_out.size(0) (search.py:385). */
int32_t wn_batch_size(const wn_model* m);

/* One-shot: the NEXT wn_encode on this handle waits for `event` (a hipEvent_t already
 * recorded by the caller, e.g. "the previous batch's encoder is finished" on another handle's
 * stream) INSIDE the call instead of the caller waiting in front of it -- behind the call's own
 * descriptor uploads and in front of GlobalCMVN + subsampling conv1 (the default, tune
 * enc_gate_pos = 0: the small host -> device copies of batch i+1 no longer stand in the chain of
 * encoders), or behind CMVN + conv1 (enc_gate_pos = 1: cmvn.py:36-47, subsampling.py:188-189,
 * the encoder's one HBM-bound kernel, then runs beside batch i's matrix-bound layers).  Used by
 * wenet_amd/pipeline.py; whatever way the call ends, the gate does not stay on the handle.
 * No reference counterpart (the reference decodes one batch at a time, recognize.py:289). */
int wn_model_set_encode_gate(wn_model* m, void* event);

/* A weight-less handle that only owns a workspace: enough for
 * wn_set_ctc_probs + the two CTC searches, i.e. for calling the reference's
 * free functions search.ctc_greedy_search / ctc_prefix_beam_search on a
 * caller-provided (B, T, V) log-prob tensor. */
int wn_workspace_create(int32_t device, wn_model** out);

/* ---- features ---------------------------------------------------------- */
/* processor.resample (dataset/processor.py:177-196): torchaudio's
 * Resample(orig_freq, new_freq) with its defaults -- polyphase windowed sinc
 * (sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99), output length
 * ceil(new * n_in / orig) = wn_resample_length().  pcm_dev (n_in) -> out_dev
 * (n_out) float, both on the device; equal rates copy. */
int64_t wn_resample_length(int64_t n_in, int32_t orig_freq, int32_t new_freq);
int wn_resample(wn_model* m, const float* pcm_dev, int64_t n_in, int32_t orig_freq,
                int32_t new_freq, float* out_dev, int64_t n_out, void* stream);

/* compute_fbank (wenet/dataset/processor.py:226-256 -> kaldi.fbank with
 * num_mel_bins, 25 ms / 10 ms, dither 0, povey window; arithmetic restated
 * from runtime/core/frontend/fbank.h:250-327) + padding
 * (processor.py:526-577, without the sort).  pcm_dev holds the B waveforms
 * back to back, float in [-1, 1]; sample_off_host has B+1 entries.
 * feats_dev is (B, max_frames, feat_dim), zero padded.
 * n_frames_host receives 1 + (n - 400) / 160 per utterance. */
int wn_fbank(wn_model* m, const float* pcm_dev, const int64_t* sample_off_host,
             int32_t B, float* feats_dev, int32_t max_frames,
             int32_t* n_frames_host, void* stream);

/* compute_log_mel_spectrogram (wenet/dataset/processor.py:320-369, the Whisper
 * frontend): hann(400) STFT with hop 160 and reflect padding, power of all
 * frames but the last, slaney mel filters (librosa.filters.mel), log10 with
 * clamp 1e-10, floor at (utterance max - 8), (x + 4) / 4.  Same buffer
 * conventions as wn_fbank; n_frames = n_samples / 160; every utterance needs
 * more than 200 samples.  `padding` / `pad_or_trim` are applied by the caller
 * to the waveform. */
int wn_log_mel(wn_model* m, const float* pcm_dev, const int64_t* sample_off_host,
               int32_t B, int32_t n_mels, float* feats_dev, int32_t max_frames,
               int32_t* n_frames_host, void* stream);

/* ---- encoder ------------------------------------------------------------ */
/* ASRModel._forward_encoder / BaseEncoder.forward (asr_model.py:216-239,
 * encoder.py:122-181): GlobalCMVN, Conv2dSubsampling4, rel-pos encoding,
 * chunk mask, n_layers x ConformerEncoderLayer, after_norm.
 * feats_dev: (B, T, feat_dim) padded; feat_lens_host: B lengths.
 * decoding_chunk_size / num_left_chunks as in the reference (<0: full).
 * enc_lens_host (B) receives the subsampled lengths.  If enc_out_dev != NULL
 * it receives the padded (B, T', d_model) output, T' = ((T-1)/2-1)/2, rows
 * past each length zero-filled (the reference leaves unspecified values
 * there).  The packed encoder output stays resident in the model's workspace
 * as the "current batch" for the wn_ctc_* / wn_rescore calls below. */
int wn_encode(wn_model* m, const float* feats_dev, const int32_t* feat_lens_host,
              int32_t B, int32_t T, int32_t decoding_chunk_size,
              int32_t num_left_chunks, float* enc_out_dev,
              int32_t* enc_lens_host, void* stream);

/* Streaming: BaseEncoder.forward_chunk (encoder.py:204-285) ==
 * ASRModel.forward_encoder_chunk (asr_model.py:385-427), batch 1, Conformer.
 *   feats_dev         (time, feat_dim) feature window of this chunk,
 *                     time = (chunk - 1) * 4 + 7 (shorter for the last chunk)
 *   offset            encoder frames emitted so far
 *   required_cache_size  < 0: keep all history; 0: none; > 0: that many frames
 *   att_cache_dev     (n_layers, heads, cache_t1, 128) K|V cache, NULL iff
 *                     cache_t1 == 0 (the reference's (0,0,0,0) tensor)
 *   cnn_cache_dev     (n_layers, 1, d_model, lorder) causal-conv left context,
 *                     NULL for the first chunk (zeros); lorder = cnn_kernel - 1
 *                     for causal models, 0 otherwise (no cnn cache at all)
 * Outputs: out_dev (chunk, d_model), chunk = ((time-1)/2-1)/2;
 * new_att_cache_dev (n_layers, heads, new_t1, 128) with
 *   new_t1 = cache_t1 + chunk - next_cache_start (encoder.py:258-263);
 * new_cnn_cache_dev like cnn_cache_dev.  The new caches must not alias the
 * inputs.  *chunk_out / *new_cache_t1_out (host, optional) return the two
 * sizes.  Nothing is synchronised; the handle's "current batch" is cleared. */
int wn_encode_chunk(wn_model* m, const float* feats_dev, int32_t time, int32_t offset,
                    int32_t required_cache_size, const float* att_cache_dev,
                    int32_t cache_t1, const float* cnn_cache_dev, float* out_dev,
                    float* new_att_cache_dev, float* new_cnn_cache_dev,
                    int32_t* chunk_out, int32_t* new_cache_t1_out, void* stream);

/* The same for n_sess streaming sessions in ONE call (the reference's
 * batched formulation is wenet/bin/export_onnx_gpu.py:83-232): feats_dev (n_sess, time,
 * feat_dim) -- every session contributes a window of the same length --, offsets_host /
 * cache_t1_host [n_sess] (sessions may be at different positions and hold caches of
 * different lengths), att_cache_dev / cnn_cache_dev / new_*_dev: host arrays of n_sess
 * device pointers, each tensor in the per-session layout above (entry NULL where that
 * session has none).  out_dev (n_sess, chunk, d_model); new_cache_t1_out [n_sess].
 * Every GEMM / LayerNorm runs on the n_sess * chunk rows at once; session b's attention
 * reads only its own [cache | chunk] keys at positions offset_b - cache_t1_b ... */
int wn_encode_chunk_batch(wn_model* m, int32_t n_sess, const float* feats_dev, int32_t time,
                          const int32_t* offsets_host, int32_t required_cache_size,
                          const float* const* att_cache_dev, const int32_t* cache_t1_host,
                          const float* const* cnn_cache_dev, float* out_dev,
                          float* const* new_att_cache_dev, float* const* new_cnn_cache_dev,
                          int32_t* chunk_out, int32_t* new_cache_t1_out, void* stream);

/* Use caller-provided padded encoder output (B, Tp, d_model) + lengths as the
 * current batch (for the reference's free functions that take encoder_out). */
int wn_set_encoder_out(wn_model* m, const float* enc_out_dev,
                       const int32_t* enc_lens_host, int32_t B, int32_t Tp,
                       void* stream);

/* ---- CTC ---------------------------------------------------------------- */
/* ASRModel.ctc_logprobs (asr_model.py:254-265): Linear(d->V) + log_softmax
 * over the current batch, with top-k per frame (k = max(1, beam)) kept in the
 * workspace for the searches.  If logp_dev != NULL the full padded
 * (B, Tp, V) log-prob tensor is also written (rows past the length zeroed). */
int wn_ctc_logprobs(wn_model* m, int32_t topk, int32_t blank_id,
                    float blank_penalty, float* logp_dev, int32_t Tp,
                    void* stream);

/* Use caller-provided padded log-probs (B, Tp, V) + lengths as the current
 * CTC posteriors (for search.ctc_greedy_search / ctc_prefix_beam_search called
 * directly on a tensor); computes the per-frame top-k without re-normalising. */
int wn_set_ctc_probs(wn_model* m, const float* logp_dev,
                     const int32_t* lens_host, int32_t B, int32_t Tp,
                     int32_t V, int32_t topk, void* stream);

/* ctc_greedy_search (search.py:109-124): tokens_host is (B, max_len) int32,
 * tok_lens_host (B). max_len must be >= the longest subsampled length. */
int wn_ctc_greedy_search(wn_model* m, int32_t blank_id, int32_t* tokens_host,
                         int32_t* tok_lens_host, int32_t max_len, void* stream);

/* Context biasing: install (n_nodes > 0) or clear (n_nodes == 0) a flattened
 * ContextGraph (wenet/utils/context_graph.py:101-265) on this handle; later
 * wn_ctc_prefix_beam_search calls are biased by it exactly like
 * ctc_prefix_beam_search(..., context_graph) (search.py:127-249): per-entry trie
 * state and bonus taken from the first contribution, second prune on
 * score + bonus, finalize() at the end.  All arrays are HOST arrays, copied.
 * Node 0 is the root (fail[0] = 0); `fail`, `node_score`, `output_score`,
 * `token_score` have n_nodes entries (ContextState.fail.id / node_score /
 * output_score / token_score); the trie arcs (ContextState.next) are the n_edges
 * triples edge_from[i] --edge_token[i]--> edge_to[i]. */
int wn_set_context_graph(wn_model* m, int32_t n_nodes, const int32_t* fail,
                         const double* node_score, const double* output_score,
                         const double* token_score, int32_t n_edges,
                         const int32_t* edge_from, const int32_t* edge_token,
                         const int32_t* edge_to, void* stream);

/* ctc_prefix_beam_search (search.py:127-249; context_graph = the graph set with
 * wn_set_context_graph, None by default).  Outputs,
 * all host: n_hyps (B); hyp_lens, hyp_tlens (B, beam); hyp_tokens, hyp_times
 * (B, beam, max_len); hyp_scores (B, beam) fp64.  beam <= 64 (<= 16 runs the
 * latency-tuned kernel).  Of every (max_len) row of hyp_tokens / hyp_times only the
 * first hyp_lens / hyp_tlens entries are written (one device -> pinned-host copy, then the
 * used corners): elements past a hypothesis' length keep what the caller put there.  The
 * n-best list also stays on the device for wn_attention_rescoring. */
int wn_ctc_prefix_beam_search(wn_model* m, int32_t beam, int32_t blank_id,
                              int32_t* n_hyps_host, int32_t* hyp_lens_host,
                              int32_t* hyp_tlens_host, int32_t* hyp_tokens_host,
                              int32_t* hyp_times_host, double* hyp_scores_host,
                              int32_t max_len, void* stream);

/* ---- attention rescoring ------------------------------------------------- */
/* attention_rescoring + ASRModel.forward_attention_decoder
 * (search.py:374-458, asr_model.py:453-547) for ALL utterances and hypotheses
 * of the current batch in one pass (the reference loops utterance by
 * utterance).  hyps given on the host as produced by the prefix beam search:
 * n_hyps (B), hyp_lens (B, beam), hyp_tokens (B, beam, max_len).
 * Outputs (host): l2r_logp, r2l_logp (B, beam, max_len + 1): log-prob of each
 * hypothesis token then of <eos> under the left-to-right decoder, and of the
 * reversed sequence under the right-to-left decoder (zeros when
 * reverse_weight == 0 or the model has no right decoder). */
int wn_attention_rescoring(wn_model* m, int32_t beam, const int32_t* n_hyps_host,
                           const int32_t* hyp_lens_host,
                           const int32_t* hyp_tokens_host, int32_t max_len,
                           float reverse_weight, float* l2r_logp_host,
                           float* r2l_logp_host, void* stream);

/* Optional, in front of wn_ctc_prefix_beam_search when wn_rescore follows on the same batch: the
 * part of attention_rescoring that does not depend on the hypotheses -- the cross-attention K / V
 * projections of the encoder output for every decoder layer (decoder_layer.py:123-138; the
 * reference recomputes them per utterance inside forward_attention_decoder) -- is queued on a
 * second stream of the handle, ordered behind `stream`'s work so far, and overlaps the search
 * (T' dependent steps on B workgroups while most CUs idle).  wn_rescore waits for it and skips
 * those projections; results are the same bits.  use_right_decoder: also for the right-to-left
 * decoder (reverse_weight > 0).  A new batch or layout (wn_encode, wn_set_encoder_out,
 * wn_filter_blank_embedding) voids the prefetch. */
int wn_rescore_prefetch(wn_model* m, int32_t use_right_decoder, void* stream);

/* attention_rescoring COMPLETE on the device (search.py:374-458; SURVEY.md 8b `wn_rescore`):
 * the decoder pass of wn_attention_rescoring, then one kernel does what search.py:424-457 does
 * in Python -- per hypothesis the fp32 left-to-right sum of the gathered log-probs + <eos>
 * (in the reference's order and dtypes), the right-to-left decoder's sum blended with
 * reverse_weight, `confidence = exp(score / (len + 1))`, `+ ctc_score * ctc_weight`, the
 * first-maximum arg-max over the hypotheses, and the winner's per-token confidences
 * (batched form in the reference: wenet/bin/export_onnx_gpu.py:666-724).
 *   n_hyps_host == NULL: the n-best is the one the LAST wn_ctc_prefix_beam_search of this
 *     handle left on the device (tokens and fp64 scores are read there; no host round trip);
 *     hyp_lens_host / hyp_tokens_host / ctc_scores_host must be NULL, beam / max_len the
 *     values that search was called with.
 *   otherwise: n_hyps (B), hyp_lens (B, beam), hyp_tokens (B, beam, max_len), ctc_scores
 *     (B, beam) host arrays (DecodeResult.nbest / nbest_scores of any source).
 * Outputs (host): best_idx (B) index into the utterance's n-best, best_score (B) fp32 (the
 * `.item()` of the reference's fp32 tensor), confidence (B) fp64, tok_conf (B, max_len) fp64
 * (first len(best hyp) entries), all_scores (B, beam) fp32 score of every hypothesis.
 * confidence / tok_conf / all_scores may be NULL.  beam <= 64. */
int wn_rescore(wn_model* m, int32_t beam, const int32_t* n_hyps_host,
               const int32_t* hyp_lens_host, const int32_t* hyp_tokens_host,
               const double* ctc_scores_host, int32_t max_len, double ctc_weight,
               double reverse_weight, int32_t* best_idx_host, float* best_score_host,
               double* confidence_host, double* tok_conf_host, float* all_scores_host,
               void* stream);

/* The decoder half of ASRModel.forward_attention_decoder (asr_model.py:453-547):
 * one decoder (`which` 0: decoder / left_decoder, 1: right_decoder,
 * decoder.py:146-201,430-463) over a PADDED batch of n_seq token rows for
 * utterance `utt` of the current batch -> log_softmax over the vocabulary of
 * every position, logp_dev (n_seq, max_len, vocab) on the device.  tokens_host
 * (n_seq, max_len) are the decoder inputs ([sos] + hyp, eos-padded); lens_host
 * the input lengths: keys past a row's length are masked, padded rows are
 * computed like the reference computes them. */
int wn_decoder_forward(wn_model* m, int32_t utt, int32_t which, int32_t n_seq,
                       const int32_t* tokens_host, const int32_t* lens_host,
                       int32_t max_len, float* logp_dev, void* stream);

/* ASRModel.filter_blank_embedding (asr_model.py:153-180), used by decode() in front of
 * attention_rescoring when model_conf.apply_non_blank_embedding is set (asr_model.py:337-342;
 * examples/aishell/s0/conf/train_u2++_lite_conformer.yaml): the encoder output of the CURRENT
 * batch is replaced by its rows whose CTC arg-max (of the posteriors wn_ctc_logprobs /
 * wn_set_ctc_probs left on the handle, blank penalty included) is not token 0, in order;
 * T = the largest number of kept rows in the batch; utterance b then has min(len_b, T) rows: its
 * kept rows followed by ZERO rows -- the reference hands attention_rescoring the zero-padded
 * (B, T, d) tensor together with the UNFILTERED lengths (search.py:396), so the decoder
 * attends to that padding too, and so does this path.  n_keep_host (B,): kept rows per
 * utterance; *t_out = T; padded_out_dev (optional, (B, T', d) or larger): the reference's
 * return tensor (B, T, d).  Only frames inside an utterance's length are considered (the
 * reference also looks at the padded frames of shorter utterances, whose encoder output is an
 * artefact of the padding; a batch of equal lengths or a single utterance is identical, and so
 * is a ragged batch none of whose padded frames comes out non-blank: tests/golden/
 * raggedlite_tiny.npz from the real reference, 1e-3.  Where padded frames ARE selected the
 * rescoring scores differ -- T and with it the number of zero rows every utterance carries
 * changes too: measured 0.10 on the random-weight case tests/golden/raggedlite_tiny_padded.npz,
 * same winners; tolerance 0.15 stated in tests/test_gpu_parity.py).  T == 0 (no non-blank frame
 * in the whole batch; the reference raises): nothing is changed, *t_out = 0, a warning is
 * printed once.  One host round trip (the kept-row counts). */
int wn_filter_blank_embedding(wn_model* m, float* padded_out_dev, int32_t* n_keep_host,
                              int32_t* t_out, void* stream);

/* attention_beam_search (search.py:252-371, the non-Whisper branch) over the CURRENT batch
 * (its encoder output is on the device after wn_encode / wn_set_encoder_out): B x beam
 * running hypotheses, one decoder row per hypothesis and step
 * (TransformerDecoder.forward_one_step, decoder.py:226-281, with a self-attention K/V
 * cache and the cross-attention K/V projected once), mask_finished_scores / _preds, the
 * N x N -> N re-ranking, the end flags and the final length-penalised arg-max all run on
 * the device; scores are fp32 like the reference's.  maxlen = the reference's
 * encoder_out.size(1).  tokens_host is (B, maxlen) int32 (the winner without <sos> /
 * <eos>), lens_host (B,).  beam <= 64.  The cache holds the steps actually run (it starts
 * at 32 steps and doubles), and the "all hypotheses ended" counter is read back every 4th
 * step: steps past the end only append <eos> to finished hypotheses, which the result strips. */
int wn_attention_beam_search(wn_model* m, int32_t beam, int32_t maxlen, float length_penalty,
                             int32_t* tokens_host, int32_t* lens_host, void* stream);

/* One step of attention_beam_search (search.py:252-371): for every running
 * hypothesis (its utterance index in the current batch, its tokens so far
 * starting with <sos>), log_softmax(output_layer(after_norm(decoder(...)[:, -1])))
 * -- TransformerDecoder.forward_one_step, decoder.py:226-281 -- reduced to its
 * `topk` best (log-prob, token) pairs, sorted.  The cross-attention K/V of the
 * current batch are projected once and kept for the following steps; the self
 * attention is recomputed over the prefix (no self-attention cache).
 * tokens_host is (n_seq, max_len) int32; outputs are (n_seq, topk), host. */
int wn_decoder_next_topk(wn_model* m, int32_t n_seq, const int32_t* seq_utt_host,
                         const int32_t* seq_lens_host, const int32_t* tokens_host,
                         int32_t max_len, int32_t topk, float* logp_host,
                         int32_t* idx_host, void* stream);

/* ---- raw operators (used by the parity tests and by other hosts) ---------- */
/* C[M,N] = resid + alpha * act(A[M,K] * W[N,K]^T + bias); act: 0 none,
 * 1 SiLU, 2 ReLU.  fp32 on v_mfma_f32_32x32x2_f32. */
int wn_op_gemm(const float* A_dev, const float* W_dev, const float* bias_dev,
               const float* resid_dev, float* C_dev, int32_t M, int32_t N,
               int32_t K, float alpha, int32_t act, void* stream);
/* The same contraction with both operands rounded to bf16 (WN_PREC_BF16). */
int wn_op_gemm_bf16(const float* A_dev, const float* W_dev, const float* bias_dev,
                    const float* resid_dev, float* C_dev, int32_t M, int32_t N,
                    int32_t K, float alpha, int32_t act, void* stream);
/* The bf16-STORAGE form of that contraction (the kernel the bf16 mode runs on
 * LayerNorm outputs / FFN hidden / attention context, which it keeps as bf16 in
 * HBM): A and W are rounded to bf16 images first, the GEMM reads those; C is fp32
 * (M, N) or, with c_bf16 != 0 and no residual, a bf16 (M, N) matrix. */
int wn_op_gemm_bf16_stored(const float* A_dev, const float* W_dev, const float* bias_dev,
                           const float* resid_dev, void* C_dev, int32_t M, int32_t N,
                           int32_t K, float alpha, int32_t act, int32_t c_bf16,
                           void* stream);
/* The reduced-precision GEMM kernels on operands that are ALREADY in their storage
 * type in HBM (what the model path hands them): dtype 1 = bf16 A (M, K) and W (N, K);
 * dtype 2 = OCP MXFP8: e4m3 A and W with one E8M0 block scale per 32 consecutive k of a
 * row, scales as dwords [K/128][pitch = M resp. N] (the 4 bytes of a dword = the 4 k
 * blocks of one 128-wide K tile), exactly what wn_op_mx_quantize writes.  c_mode 0: fp32
 * C (M, N); 1: bf16 C (bf16 operands, no residual); 2: MXFP8 C (MXFP8 operands, no
 * residual) with its block scales [N/128][M] in c_scale_dev.  Micro-benchmarks and the
 * operator tests call this. */
int wn_op_gemm_lowp(const void* A_dev, const void* W_dev, const void* a_scale_dev,
                    const void* w_scale_dev, const float* bias_dev,
                    const float* resid_dev, void* C_dev, void* c_scale_dev, int32_t M,
                    int32_t N, int32_t K, float alpha, int32_t act, int32_t c_mode,
                    int32_t dtype, void* stream);
/* x (rows, K) fp32 -> MXFP8: q (rows, K) e4m3 bytes + scale dwords [K/128][rows]
 * (block rule: the smallest power of two 2^e with amax <= 448 * 2^e; csrc/mxfp8.h). */
int wn_op_mx_quantize(const float* x_dev, int32_t rows, int32_t K, void* q_dev,
                      void* scale_dev, void* stream);
/* The fused fp32 feed-forward module the encoder runs (csrc/ffn_fused.hip;
 * positionwise_feed_forward.py:50-58 inside encoder_layer.py:220-228): x_inout (M, D) +=
 * alpha * (act(X W1^T + b1) W2^T + b2) with X (M, D) the already normalised input, then
 * y_out = LayerNorm(x_inout; ln_w, ln_b, eps).  D in {256, 512}, F % 64 == 0, act 1 SiLU /
 * 2 ReLU / 3 GELU.  Operator tests and micro-benchmarks call this. */
int wn_op_ffn_fused(const float* X_dev, const float* W1_dev, const float* b1_dev,
                    const float* W2_dev, const float* b2_dev, float* x_inout_dev,
                    const float* ln_w_dev, const float* ln_b_dev, float* y_out_dev,
                    int32_t M, int32_t D, int32_t F, int32_t act, float alpha, float eps,
                    void* stream);
/* fp32 GEMM on the bf16 matrix cores (csrc/gemm_x6.hip): every fp32 operand is split
 * exactly into three bf16 planes and six of the nine plane products are accumulated in
 * fp32 (the dropped ones are below 2^-26 of a product).  C (M, N) = resid + alpha *
 * act(A W^T + bias); A (M, K), W (N, K) fp32 row-major, K % 16 == 0, N % 4 == 0.  `bm`:
 * block rows 128 / 256 (0 = auto); `reps` > 1 repeats the GEMM launch (micro-benchmarks).
 * Replaces torch.nn.functional.linear as the reference's layers call it
 * (positionwise_feed_forward.py:50-58, attention.py:100-176, convolution.py:120-148). */
int wn_op_gemm_x6(const float* A_dev, const float* W_dev, const float* bias_dev,
                  const float* resid_dev, float* C_dev, int32_t M, int32_t N, int32_t K,
                  float alpha, int32_t act, int32_t bm, int32_t reps, void* stream);
/* wn_op_ffn_fused's contract with both contractions as six bf16 plane products: by default
 * (d_model 256, SiLU / ReLU) ONE launch of csrc/ffn_x6f.hip with the hidden tensor in
 * registers; wn_tune_set("ffn_x6f", 0) or any other shape: two launches of the six-product
 * GEMM, the hidden tensor going from the first one's epilogue to the second as a plane image. */
int wn_op_ffn_x6(const float* X_dev, const float* W1_dev, const float* b1_dev,
                 const float* W2_dev, const float* b2_dev, float* x_inout_dev,
                 const float* ln_w_dev, const float* ln_b_dev, float* y_out_dev, int32_t M,
                 int32_t D, int32_t F, int32_t act, float alpha, float eps, int32_t reps,
                 void* stream);
/* Row-block six-product GEMM with the A rows in registers (csrc/gemm_x6r.hip), K = 256:
 * epi 0: C (M, N) = A W^T + bias, N in {256, 512, 768} (the QKV projection,
 * attention.py:109-131); epi 1 (N = 256): x <- x + alpha (A W^T + bias), y <- LayerNorm(x; ln_w,
 * ln_b, eps) -- the attention output projection / pointwise_conv2 with the residual and the
 * LayerNorm that follows (encoder_layer.py:238-240, 251-253).  `reps` launches (micro-benchmark;
 * epi 1 then accumulates into x every time). */
int wn_op_gemm_x6r(const float* A_dev, const float* W_dev, const float* bias_dev,
                   float* x_inout_dev, const float* ln_w_dev, const float* ln_b_dev,
                   float* y_dev, float* C_dev, int32_t M, int32_t N, int32_t epi, float alpha,
                   float eps, int32_t reps, void* stream);

/* The d_model = 512 row-block kernels (csrc/gemm_x6r512.hip), test / benchmark entry: K = 512;
 * epi 0: C (M, N) = A W^T + bias, N in {512, 1024, 1536, 2048}; epi 1: x_inout += alpha (A W^T +
 * bias), y = LayerNorm(x_inout), N = 512; epi 3: epi 1, then C (M, 512) = GLU(y W2^T + bias2)
 * with W2 (1024, 512) in the [32 values | 32 gates]-per-64 row order wn_model_create gives
 * pointwise_conv1 (y is written only if non-null). */
int wn_op_gemm_x6r512(const float* A, const float* W, const float* bias, float* x_inout,
                      const float* ln_w, const float* ln_b, float* y, const float* W2,
                      const float* bias2, float* C, int32_t M, int32_t N, int32_t epi, float alpha,
                      float eps, int32_t reps, void* stream);
/* out[i] = log_add(a[i], b[i]) (wenet/utils/common.py:302-310) in fp64 with the
 * routine the prefix beam search uses. */
int wn_op_log_add(const double* a_dev, const double* b_dev, double* out_dev,
                  int32_t n, void* stream);
int wn_op_layernorm(const float* x_dev, const float* w_dev, const float* b_dev,
                    float* y_dev, int32_t M, int32_t D, float eps, void* stream);

/* Measurement hook for bench.py: bracket launches of the dominant kernel (the
 * FFN w_1 GEMM, positionwise_feed_forward.py:58; every 6th launch, because each
 * event pair idles the GPU for ~10 us) with HIP events on the launch stream.  wn_profile_collect waits for them and returns the number of
 * launches, their summed duration and their summed algorithmic FLOPs
 * (2*M*N*K each) since the last enable/collect.  on = 1: every 6th launch; on = N > 1: every
 * N-th (bench.py's timed rounds use one bracket per decode, its single-stream roofline pass 6). */
int wn_profile_enable(wn_model* m, int32_t on);
int wn_profile_collect(wn_model* m, int32_t* n_launches, double* total_ms,
                       double* total_flops);
/* What the bracketed launches were (static string): the FFN w_1 GEMM -- in the fp32 mode the
 * fused six-product kernel of csrc/ffn_x6f.hip (w_1 + activation + w_2, d_model 256) or the
 * six-product w_1 GEMM of csrc/gemm_x6.hip -- or the fused v_mfma_f32 feed-forward kernel when
 * wn_tune_set("gemm_x6", 0) puts the fp32 path on it. */
const char* wn_profile_kernel_name(const wn_model* m);
/* Hidden slices S (fused forms: partial sums [S][M][d] reduced by the next kernel) or K slices
 * of the w_2 GEMM of the feed-forward module this handle ran last -- what bench.py prices the
 * algorithmic bytes of the roofline kernel with (positionwise_feed_forward.py:50-58 has no such
 * notion: it is a property of the launch geometry). */
int32_t wn_profile_ffn_split(const wn_model* m);
/* Measurement: shader-clock stamps [4 waves][24] written by the clock-stamp variants of the fused
 * feed-forward kernel (wn_tune_set("ffn_x6f_var", 8704 ...), tools/bench_x6.py --clocks): entry
 * i = start of sub-stage i of one block's last steady-state chunk, entry 8 = its end. */
int wn_profile_ffn_clocks(uint64_t* out64);
/* Same for the six-product tile GEMM (wn_tune_set("x6_probe", 4)): [8 waves][8] = entry, loop
 * start, loop end, kernel end (shader clock), 100-MHz real time at entry / end, k blocks.  While
 * wn_tune_set("lp_probe", 4 [| 8 first block | 16 last block]) is set, the stamps of the pipelined
 * bf16 / MXFP8 GEMM instead: entry, first tile landed, K loop end, last store issued, stores
 * drained, real time at entry / end, K tiles (tools/lp_clocks.py).  With "x6_probe" = 8: the
 * five phase stamps block 0 of the last row-block GEMM launch (csrc/gemm_x6r.hip) left: entry,
 * rows in LDS, planes in LDS, main loop done, stores drained (tools/x6r_clocks.py). */
int wn_profile_gemm_clocks(uint64_t* out64);

/* Test hook: "n_layers" = run only the first n encoder layers (-1: all),
 * "skip_after_norm" = 1 leaves out encoder.after_norm; lets the parity tests
 * compare every ConformerEncoderLayer output with the oracle. */
int wn_debug_set(wn_model* m, const char* key, int32_t value);

/* Tuning knobs: A/B switches between kernel forms, measurement probes and test hooks.  The key
 * list, what every value selects and the defaults (= the shipped configuration) are ONE table,
 * WN_TUNE_KEYS in wenet_amd/csrc/tune.h; unknown keys are an error.  Values that are wrong by
 * design (ablations) are refused unless the library was built with WN_ABLATION=1.
 *
 * wn_tune_set writes the PROCESS DEFAULT: what handle-less operators (wn_op_*) and every handle
 * without an override of its own see (tools/bench_*.py, bench.py --tune, most tests).
 * wn_model_tune_set writes ONE HANDLE's override (INT32_MIN = drop the override, follow the process
 * default again); a handle's entry points run on its effective set, resolved when the call enters
 * and current only for the calling thread, so two handles driven by two host threads can run
 * different kernel forms side by side.  wn_model_clone copies the overrides.  wn_tune_get reads
 * the effective value for a handle (NULL: the process default).
 * No counterpart in the reference: its kernels are chosen by torch's dispatcher. */
int wn_tune_set(const char* key, int32_t value);
int wn_model_tune_set(wn_model* m, const char* key, int32_t value);
int wn_tune_get(const wn_model* m, const char* key, int32_t* value);

#ifdef __cplusplus
}
#endif
#endif /* WENET_AMD_H_ */
