"""ctypes binding of libwenet_amd.so (include/wenet_amd.h).

There is deliberately no fallback: if the HIP library is missing or fails to
load, every use of the model raises.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int32,
                    c_int64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libwenet_amd.so')


class WnConfig(Structure):
    _fields_ = [(n, c_int32) for n in (
        'feat_dim', 'd_model', 'n_heads', 'ffn_dim', 'n_layers', 'cnn_kernel',
        'causal', 'use_dynamic_chunk', 'static_chunk_size', 'vocab',
        'has_cmvn', 'dec_heads', 'dec_ffn_dim', 'dec_layers', 'dec_r_layers',
        'bidirectional', 'sos', 'eos', 'max_pos')] + [('norm_eps', c_float)] + [
            (n, c_int32) for n in ('encoder_type', 'input_layer', 'activation',
                                   'key_bias', 'cnn_norm')]


class WnTensor(Structure):
    _fields_ = [('name', c_char_p), ('data', POINTER(c_float)),
                ('numel', c_int64)]


# every symbol include/wenet_amd.h declares
EXPORTS = [
    'wn_last_error', 'wn_version', 'wn_model_create', 'wn_model_destroy', 'wn_model_clone',
    'wn_model_set_precision', 'wn_model_get_precision', 'wn_batch_size', 'wn_model_set_encode_gate', 'wn_op_gemm_bf16',
    'wn_op_gemm_bf16_stored',
    'wn_attention_beam_search', 'wn_encode_chunk_batch', 'wn_op_gemm_lowp', 'wn_op_mx_quantize', 'wn_op_ffn_fused', 'wn_op_gemm_x6', 'wn_op_ffn_x6', 'wn_op_gemm_x6r', 'wn_op_gemm_x6r512', 'wn_profile_kernel_name', 'wn_profile_ffn_split', 'wn_profile_ffn_clocks', 'wn_profile_gemm_clocks', 'wn_filter_blank_embedding',
    'wn_workspace_create', 'wn_resample_length', 'wn_resample', 'wn_fbank', 'wn_log_mel', 'wn_encode', 'wn_encode_chunk', 'wn_set_encoder_out',
    'wn_ctc_logprobs', 'wn_set_ctc_probs', 'wn_ctc_greedy_search',
    'wn_set_context_graph', 'wn_ctc_prefix_beam_search', 'wn_attention_rescoring', 'wn_rescore', 'wn_rescore_prefetch', 'wn_decoder_forward', 'wn_decoder_next_topk', 'wn_op_gemm',
    'wn_op_layernorm', 'wn_op_log_add', 'wn_debug_set', 'wn_profile_enable',
    'wn_profile_collect', 'wn_tune_set', 'wn_model_tune_set', 'wn_tune_get',
]

_lib = None


def lib():
    """Load (once) and return the library; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} is missing: build it with `python -m wenet_amd.build` '
            '(hipcc --offload-arch=gfx950). wenet_amd has no CPU fallback.')
    # torch first: its bundled libamdhip64.so.7 must be the one HIP runtime in
    # the process so that torch device pointers and ours share a context.
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    i32, i64, f32, vp = c_int32, c_int64, c_float, c_void_p
    pi32, pf64 = POINTER(c_int32), POINTER(c_double)
    L.wn_last_error.restype = c_char_p
    L.wn_profile_kernel_name.restype = c_char_p
    L.wn_profile_kernel_name.argtypes = [vp]
    L.wn_profile_ffn_split.restype = c_int32
    L.wn_profile_ffn_split.argtypes = [vp]
    L.wn_profile_ffn_clocks.argtypes = [POINTER(ctypes.c_uint64)]
    L.wn_profile_gemm_clocks.argtypes = [POINTER(ctypes.c_uint64)]
    L.wn_version.restype = c_char_p
    L.wn_model_create.argtypes = [POINTER(WnConfig), POINTER(WnTensor), i32,
                                  i32, POINTER(vp)]
    L.wn_model_destroy.argtypes = [vp]
    L.wn_model_clone.argtypes = [vp, POINTER(vp)]
    L.wn_model_destroy.restype = None
    L.wn_workspace_create.argtypes = [i32, POINTER(vp)]
    L.wn_fbank.argtypes = [vp, vp, POINTER(i64), i32, vp, i32, pi32, vp]
    L.wn_log_mel.argtypes = [vp, vp, POINTER(i64), i32, i32, vp, i32, pi32, vp]
    L.wn_encode.argtypes = [vp, vp, pi32, i32, i32, i32, i32, vp, pi32, vp]
    L.wn_set_encoder_out.argtypes = [vp, vp, pi32, i32, i32, vp]
    L.wn_ctc_logprobs.argtypes = [vp, i32, i32, f32, vp, i32, vp]
    L.wn_encode_chunk.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp, vp, vp, vp, pi32,
                                  pi32, vp]
    L.wn_encode_chunk_batch.argtypes = [vp, i32, vp, i32, pi32, i32, POINTER(vp), pi32,
                                        POINTER(vp), vp, POINTER(vp), POINTER(vp), pi32, pi32,
                                        vp]
    L.wn_resample_length.argtypes = [ctypes.c_int64, i32, i32]
    L.wn_resample_length.restype = ctypes.c_int64
    L.wn_resample.argtypes = [vp, vp, ctypes.c_int64, i32, i32, vp, ctypes.c_int64, vp]
    L.wn_set_ctc_probs.argtypes = [vp, vp, pi32, i32, i32, i32, i32, vp]
    L.wn_ctc_greedy_search.argtypes = [vp, i32, pi32, pi32, i32, vp]
    L.wn_set_context_graph.argtypes = [vp, i32, pi32, pf64, pf64, pf64, i32, pi32, pi32,
                                       pi32, vp]
    L.wn_ctc_prefix_beam_search.argtypes = [vp, i32, i32, pi32, pi32, pi32,
                                            pi32, pi32, pf64, i32, vp]
    L.wn_attention_rescoring.argtypes = [vp, i32, pi32, pi32, pi32, i32, f32,
                                         POINTER(f32), POINTER(f32), vp]
    L.wn_rescore.argtypes = [vp, i32, pi32, pi32, pi32, pf64, i32, c_double, c_double, pi32,
                             POINTER(f32), pf64, pf64, POINTER(f32), vp]
    L.wn_rescore_prefetch.argtypes = [vp, i32, vp]
    L.wn_decoder_forward.argtypes = [vp, i32, i32, i32, pi32, pi32, i32, vp, vp]
    L.wn_decoder_next_topk.argtypes = [vp, i32, pi32, pi32, pi32, i32, i32,
                                       POINTER(f32), pi32, vp]
    L.wn_attention_beam_search.argtypes = [vp, i32, i32, f32, pi32, pi32, vp]
    L.wn_op_gemm.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp]
    L.wn_op_gemm_bf16.argtypes = L.wn_op_gemm.argtypes
    L.wn_op_gemm_bf16_stored.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, i32, vp]
    L.wn_op_gemm_lowp.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, i32,
                                  i32, vp]
    L.wn_op_mx_quantize.argtypes = [vp, i32, i32, vp, vp, vp]
    L.wn_op_ffn_fused.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32,
                                  f32, vp]
    L.wn_op_gemm_x6.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, i32, i32, vp]
    L.wn_op_ffn_x6.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32,
                               f32, i32, vp]
    L.wn_filter_blank_embedding.argtypes = [vp, vp, pi32, pi32, vp]
    L.wn_op_gemm_x6r.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, i32, vp]
    L.wn_op_gemm_x6r512.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32,
                                    i32, vp]
    L.wn_model_set_precision.argtypes = [vp, i32]
    L.wn_model_get_precision.argtypes = [vp]
    L.wn_batch_size.argtypes = [vp]
    L.wn_batch_size.restype = i32
    L.wn_model_set_encode_gate.argtypes = [vp, vp]
    L.wn_op_layernorm.argtypes = [vp, vp, vp, vp, i32, i32, f32, vp]
    L.wn_op_log_add.argtypes = [vp, vp, vp, i32, vp]
    L.wn_debug_set.argtypes = [vp, c_char_p, i32]
    L.wn_tune_set.argtypes = [c_char_p, i32]
    L.wn_model_tune_set.argtypes = [vp, c_char_p, i32]
    L.wn_tune_get.argtypes = [vp, c_char_p, pi32]
    L.wn_profile_enable.argtypes = [vp, i32]
    L.wn_profile_collect.argtypes = [vp, pi32, pf64, pf64]
    for n in EXPORTS:
        if n not in ('wn_last_error', 'wn_version', 'wn_model_destroy',
                     'wn_resample_length', 'wn_profile_kernel_name',
                     'wn_profile_ffn_split'):
            getattr(L, n).restype = i32
    _lib = L
    return L


def check(status: int, what: str = ''):
    if status != 0:
        msg = lib().wn_last_error().decode('utf8', 'replace')
        if status == -4:  # handle busy (one host thread per handle)
            raise RuntimeError(f'{what}: {msg}')
        if status == -1 and ('must not be 0' in msg or 'null' in msg):
            raise AssertionError(f'{what}: {msg}')
        raise RuntimeError(f'{what} failed ({status}): {msg}')


def i32p(a):
    return a.ctypes.data_as(POINTER(c_int32))


def f32p(a):
    return a.ctypes.data_as(POINTER(c_float))


def f64p(a):
    return a.ctypes.data_as(POINTER(c_double))


def i64p(a):
    return a.ctypes.data_as(POINTER(c_int64))
