"""`ASRModel` / `load_model` with the reference's API surface, running on
libwenet_amd's HIP kernels.

Mirrors (reference file:line):
  * wenet.load_model(model_dir, device)          wenet/cli/model.py:71-110
  * ASRModel.decode(methods, speech, lens, ...)  wenet/models/transformer/asr_model.py:267-343
  * ASRModel.transcribe(wav)                     asr_model.py:345-358
  * ASRModel._forward_encoder / ctc_logprobs     asr_model.py:216-239, 254-265
  * sos_symbol / eos_symbol / subsampling_rate / right_context /
    is_bidirectional_decoder                     asr_model.py:360-383, 443-451

torch is used only as the tensor container (device memory, streams) and to read
`final.pt`.  There is no CPU path: everything below raises if the HIP library
is missing or the tensors are not on a GPU.
"""
import ctypes
import json
import math
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from wenet_amd import _lib
from wenet_amd.search import (DecodeResult, _greedy, _prefix_beam,
                              _nbest_arrays, _require_cuda, _stream_ptr,
                              rescore_nbest)

_SUPPORTED = ('ctc_greedy_search', 'ctc_prefix_beam_search',
              'attention_rescoring')


def config_from_yaml(configs: dict) -> _lib.WnConfig:
    """train.yaml dict -> wn_config (keys as init_model.py:100-181 reads them)."""
    ec = configs['encoder_conf']
    dc = configs.get('decoder_conf') or {}
    # init_model.py:36-50 WENET_MODEL_CLASSES: the CTC / attention-decoder family
    # (ASRModel and its subclasses) is on this path; transducer, paraformer, ...
    # have other decoders and decode methods.
    model_type = configs.get('model', 'asr_model')
    if model_type not in ('asr_model', 'whisper', 'k2_model', 'ctl_model'):
        raise NotImplementedError(
            f'model {model_type!r} is outside the accelerated path (asr_model family)')
    enc_type = configs.get('encoder', 'conformer')
    if enc_type == 'conformer':
        checks = dict(input_layer='conv2d', pos_enc_layer_type='rel_pos',
                      selfattention_layer_type='rel_selfattn',
                      activation_type='swish',
                      normalize_before=True, use_cnn_module=True,
                      macaron_style=True)
        defaults = dict(input_layer='conv2d', pos_enc_layer_type='rel_pos',
                        selfattention_layer_type='rel_selfattn',
                        activation_type='swish', cnn_module_norm='batch_norm',
                        normalize_before=True, use_cnn_module=True,
                        macaron_style=True)
    elif enc_type == 'transformer':
        # TransformerEncoder as the Whisper recipes configure it
        # (examples/aishell/whisper/conf/finetune_whisper_largev3.yaml:1-17)
        checks = dict(input_layer='conv1d2', pos_enc_layer_type='abs_pos_whisper',
                      activation_type='gelu', normalize_before=True,
                      selfattention_layer_type='selfattn',
                      layer_norm_type='layer_norm',
                      mlp_type='position_wise_feed_forward')
        defaults = dict(input_layer='conv2d', pos_enc_layer_type='abs_pos',
                        activation_type='relu', normalize_before=True,
                        selfattention_layer_type='selfattn',
                        layer_norm_type='layer_norm',
                        mlp_type='position_wise_feed_forward')
    else:
        raise NotImplementedError(
            f'encoder {enc_type!r} is outside the accelerated path')
    for k, v in checks.items():
        got = ec.get(k, defaults[k])
        if got != v:
            raise NotImplementedError(
                f'encoder_conf.{k}={got!r} is outside the accelerated path '
                f'(needs {v!r})')
    # Keys that change the architecture and have no kernel here: any value other
    # than the reference default (encoder.py:44-63,383-400,464-483, decoder.py:59-84)
    # is refused instead of being decoded with the wrong network.
    enc_only_default = dict(layer_norm_type='layer_norm', mlp_type='position_wise_feed_forward',
                            final_norm=True, query_bias=True, value_bias=True, mlp_bias=True,
                            n_kv_head=None, head_dim=None, n_expert=8,
                            n_expert_activated=2, gradient_checkpointing=False,
                            use_sdpa=False, positionwise_conv_kernel_size=1)
    if enc_type == 'conformer':
        enc_only_default.update(conv_bias=True, conv_inner_factor=2, key_bias=True,
                                conv_norm_eps=ec.get('norm_eps', 1e-5))
    for k, v in enc_only_default.items():
        if k in ec and ec[k] != v and not (k in ('use_sdpa', 'gradient_checkpointing')):
            raise NotImplementedError(
                f'encoder_conf.{k}={ec[k]!r} is outside the accelerated path '
                f'(only the reference default {v!r} has kernels)')
    dec_default = dict(normalize_before=True, src_attention=True, query_bias=True,
                       key_bias=True, value_bias=True, mlp_bias=True, n_kv_head=None,
                       head_dim=None, layer_norm_type='layer_norm',
                       mlp_type='position_wise_feed_forward', tie_word_embedding=False,
                       use_output_layer=True)
    if enc_type == 'conformer':
        dec_default.update(activation_type='relu', input_layer='embed')
    dec_off = False   # Whisper-style decoders: the encoder (+ CTC head) runs, not the decoder
    for k, v in dec_default.items():
        if k in dc and dc[k] != v:
            if enc_type == 'transformer':
                dec_off = True
                continue
            raise NotImplementedError(
                f'decoder_conf.{k}={dc[k]!r} is outside the accelerated path '
                f'(only the reference default {v!r} has kernels)')
    vocab = configs['output_dim']
    st = (configs.get('tokenizer_conf') or {}).get('special_tokens')
    if model_type == 'whisper':
        # Whisper.__init__ (models/whisper/whisper.py:45-52): sos / eos are the
        # tokenizer's 'sot' / 'eot'
        sos = vocab - 1 if st is None else st.get('sot', vocab - 1)
        eos = vocab - 1 if st is None else st.get('eot', vocab - 1)
    else:
        sos = vocab - 1 if st is None else st.get('<sos>', vocab - 1)
        eos = vocab - 1 if st is None else st.get('<eos>', vocab - 1)
    dec_type = configs.get('decoder', 'bitransformer')
    bidir = dec_type == 'bitransformer'
    c = _lib.WnConfig()
    c.feat_dim = configs['input_dim']
    c.d_model = ec.get('output_size', 256)
    c.n_heads = ec.get('attention_heads', 4)
    c.ffn_dim = ec.get('linear_units', 2048)
    c.n_layers = ec.get('num_blocks', 6)
    c.cnn_kernel = ec.get('cnn_module_kernel', 15)
    c.causal = int(bool(ec.get('causal', False)))
    c.use_dynamic_chunk = int(bool(ec.get('use_dynamic_chunk', False)))
    c.static_chunk_size = ec.get('static_chunk_size', 0)
    c.vocab = vocab
    c.has_cmvn = int(configs.get('cmvn', None) == 'global_cmvn')
    c.dec_heads = dc.get('attention_heads', 4)
    c.dec_ffn_dim = dc.get('linear_units', 2048)
    c.dec_layers = dc.get('num_blocks', 6) if dec_type is not None else 0
    c.dec_r_layers = dc.get('r_num_blocks', 0) if bidir else 0
    c.bidirectional = int(bidir)
    c.sos, c.eos = sos, eos
    c.max_pos = 5000
    c.norm_eps = ec.get('norm_eps', 1e-5)
    c.encoder_type = 1 if enc_type == 'transformer' else 0
    c.input_layer = 1 if ec.get('input_layer') == 'conv1d2' else 0
    c.activation = 1 if ec.get('activation_type', 'swish') == 'gelu' else 0
    c.key_bias = int(bool(ec.get('key_bias', True)))
    cnn_norm = ec.get('cnn_module_norm', 'batch_norm')  # encoder.py default
    if enc_type == 'conformer' and cnn_norm not in ('layer_norm', 'batch_norm'):
        raise NotImplementedError(
            f'encoder_conf.cnn_module_norm={cnn_norm!r} is outside the accelerated path')
    c.cnn_norm = 1 if (enc_type == 'conformer' and cnn_norm == 'batch_norm') else 0
    if enc_type == 'transformer':
        c.cnn_kernel, c.causal = 1, 0
        # the Whisper decoder (learnable positions, tied embedding) is not on
        # the accelerated path: encoder (+ CTC head) only
        if dec_off or dc.get('input_layer', 'embed') != 'embed' or \
                dc.get('activation_type', 'relu') != 'relu':
            c.dec_layers = c.dec_r_layers = 0
            c.bidirectional = 0
    return c


def load_cmvn(cmvn_file: str, is_json: bool):
    """wenet/utils/cmvn.py:21-93 -> (mean, istd) float64 arrays."""
    if is_json:
        with open(cmvn_file) as f:
            st = json.load(f)
        means, variance, count = st['mean_stat'], st['var_stat'], st['frame_num']
    else:
        with open(cmvn_file) as f:
            arr = f.read().split()
        assert arr[0] == '[' and arr[-2] == '0' and arr[-1] == ']'
        dim = int((len(arr) - 2 - 2) / 2)
        means = [float(x) for x in arr[1:dim + 1]]
        count = float(arr[dim + 1])
        variance = [float(x) for x in arr[dim + 2:2 * dim + 2]]
    means = list(means)
    variance = list(variance)
    for i in range(len(means)):
        means[i] /= count
        variance[i] = variance[i] / count - means[i] * means[i]
        if variance[i] < 1.0e-20:
            variance[i] = 1.0e-20
        variance[i] = 1.0 / math.sqrt(variance[i])
    return np.array(means), np.array(variance)


class _Tokenizer:
    """id -> symbol table from units.txt; `detokenize` like
    wenet/text/base_tokenizer.py:14 + char/bpe tokenizers (join the symbols;
    sentencepiece's word-boundary mark becomes a space)."""

    bpe_path = None

    def __init__(self, units_file: str):
        self.id2sym = {}
        with open(units_file, 'r', encoding='utf8') as f:
            for line in f:
                arr = line.strip().split()
                if len(arr) == 2:
                    self.id2sym[int(arr[1])] = arr[0]
        self.symbol_table = {s: i for i, s in self.id2sym.items()}

    def detokenize(self, ids: List[int]) -> Tuple[str, List[str]]:
        tokens = [self.id2sym.get(int(i), '<unk>') for i in ids]
        text = ''.join(tokens).replace('▁', ' ').strip()
        return text, tokens


def reverse_hyps(hyps: torch.Tensor, hyps_lens: torch.Tensor, eos: int) -> torch.Tensor:
    """Decoder input of the right-to-left decoder (asr_model.py:485-536): `hyps`
    (N, L) start with sos and are eos-padded, `hyps_lens` count the sos; each
    hypothesis is reversed behind its sos and padded with eos again."""
    r_lens = hyps_lens - 1
    r_hyps = hyps[:, 1:]
    max_len = int(r_lens.max())
    idx_range = torch.arange(0, max_len)
    seq_mask = r_lens.unsqueeze(1) > idx_range
    index = ((r_lens.unsqueeze(1) - 1) - idx_range) * seq_mask
    r_hyps = torch.gather(r_hyps, 1, index)
    r_hyps = torch.where(seq_mask, r_hyps, torch.tensor(eos))
    return torch.cat([hyps[:, 0:1], r_hyps], dim=1)


class ASRModel:
    """GPU-resident Conformer CTC/attention model with the reference's
    inference API."""

    default_decode_method = 'attention_rescoring'  # asr_model.py:40

    def __init__(self, configs: dict, state_dict: Dict[str, torch.Tensor],
                 device='cuda'):
        self.configs = configs
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError(
                'wenet_amd.ASRModel needs a GPU device (MI355X); there is no '
                'CPU fallback -- use the reference for CPU decoding')
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self._cfg = config_from_yaml(configs)
        pe = state_dict.get('encoder.embed.pos_enc.pe')
        if pe is not None:  # WhisperPositionalEncoding keeps 1500 rows
            self._cfg.max_pos = int(pe.shape[-2])
        self.vocab_size = self._cfg.vocab
        self.sos, self.eos = self._cfg.sos, self._cfg.eos
        self.ignore_id = -1
        mc = configs.get('model_conf') or {}
        self.ctc_weight = mc.get('ctc_weight', 0.5)
        self.reverse_weight = mc.get('reverse_weight', 0.0)
        # asr_model.py:46,337-342: attention_rescoring sees only the non-blank frames
        self.apply_non_blank_embedding = bool(mc.get('apply_non_blank_embedding', False))
        # True when the last attention_rescoring decode found no non-blank frame in the whole
        # batch and rescored against the unfiltered encoder output (deviations);
        # reset by every decode, copied back to the caller's model by DecodePipeline
        self.last_non_blank_filter_empty = False
        self.special_tokens = (configs.get('tokenizer_conf') or {}).get(
            'special_tokens')
        L = _lib.lib()
        arrs, tensors = [], (_lib.WnTensor * len(state_dict))()
        for i, (k, v) in enumerate(state_dict.items()):
            a = np.ascontiguousarray(
                v.detach().to('cpu', torch.float32).numpy()
                if isinstance(v, torch.Tensor) else np.asarray(v, np.float32))
            arrs.append(a)
            tensors[i].name = k.encode('utf8')
            tensors[i].data = _lib.f32p(a)
            tensors[i].numel = a.size
        h = ctypes.c_void_p()
        _lib.check(
            L.wn_model_create(ctypes.byref(self._cfg), tensors,
                              len(state_dict), self.device.index,
                              ctypes.byref(h)), 'wn_model_create')
        self._h = h.value
        self._L = L
        self._last_prefix_raw = None

    def clone(self) -> 'ASRModel':
        """A second model object on the SAME device weights with its own
        workspace (wn_model_clone): one per in-flight batch
        (wenet_amd/pipeline.py)."""
        other = object.__new__(ASRModel)
        other.__dict__.update({k: v for k, v in self.__dict__.items()
                               if k not in ('_h', '_last_prefix_raw')})
        h = ctypes.c_void_p()
        _lib.check(self._L.wn_model_clone(self._h, ctypes.byref(h)),
                   'wn_model_clone')
        other._h = h.value
        other._last_prefix_raw = None
        return other

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            try:
                self._L.wn_model_destroy(h)
            except Exception:
                pass
            self._h = None

    # ---- compute dtype (recognize.py --dtype, recognize.py:52-56,250-255) ----
    _DTYPES = {'fp32': 0, 'float32': 0, torch.float32: 0,
               'bf16': 1, 'bfloat16': 1, torch.bfloat16: 1,
               'fp8': 2, 'mxfp8': 2}

    def set_compute_dtype(self, dtype) -> 'ASRModel':
        """'fp32' (default, the parity mode) or 'bf16': every Linear / pointwise
        conv / subsampling conv rounds its operands to bf16 and accumulates in
        fp32 on the bf16 matrix cores (wn_model_set_precision).  The reference
        gets the same effect from torch autocast around model.decode; 'fp16' is
        not offered (no fp16 kernels).  'fp8': the bf16 mode with the encoder's
        feed-forward GEMMs on OCP MXFP8 operands (WN_PREC_FP8, BASELINE.json
        configs[4])."""
        if dtype not in self._DTYPES:
            raise ValueError(f"compute dtype must be 'fp32', 'bf16' or 'fp8', got {dtype!r}")
        _lib.check(self._L.wn_model_set_precision(self._h, self._DTYPES[dtype]),
                   'wn_model_set_precision')
        return self

    def tune(self, key: str, value=None) -> int:
        """This handle's tuning knob `key` (csrc/tune.h): with `value`, override it for this
        handle only (wn_model_tune_set; 'inherit' drops the override); returns the effective
        value.  Process-wide defaults: wn_tune_set.  No counterpart in the reference."""
        if value is not None:
            v = -2 ** 31 if value == 'inherit' else int(value)
            _lib.check(self._L.wn_model_tune_set(self._h, key.encode(), v), 'wn_model_tune_set')
        out = ctypes.c_int32(0)
        _lib.check(self._L.wn_tune_get(self._h, key.encode(), ctypes.byref(out)), 'wn_tune_get')
        return out.value

    @property
    def compute_dtype(self) -> str:
        return {1: 'bf16', 2: 'fp8'}.get(self._L.wn_model_get_precision(self._h), 'fp32')

    # torch.nn.Module-flavoured no-ops so reference driver code keeps working
    def eval(self):
        return self

    def to(self, device):
        if torch.device(device).type != 'cuda':
            raise RuntimeError('wenet_amd models live on the GPU only')
        return self

    # ---- exported symbols ------------------------------------------------
    def subsampling_rate(self) -> int:  # asr_model.py:360-365
        return 2 if self._cfg.input_layer == 1 else 4

    def right_context(self) -> int:  # asr_model.py:367-371
        # Conv1dSubsampling2: 4 (subsampling.py:143); Conv2dSubsampling4: 6
        return 4 if self._cfg.input_layer == 1 else 6

    def sos_symbol(self) -> int:
        return self.sos

    def eos_symbol(self) -> int:
        return self.eos

    def is_bidirectional_decoder(self) -> bool:
        return bool(self._cfg.bidirectional)

    # ---- stages ----------------------------------------------------------
    def _prep(self, speech: torch.Tensor, speech_lengths: torch.Tensor):
        assert speech.shape[0] == speech_lengths.shape[0]
        speech = speech.detach().to(self.device, torch.float32).contiguous()
        lens = speech_lengths.detach().cpu().numpy().astype(np.int32)
        return speech, lens

    def _encode(self, speech, lens, chunk, left, want_out: bool):
        B, T, F = speech.shape
        assert F == self._cfg.feat_dim, 'feature dimension mismatch'
        if self._cfg.input_layer == 1:  # Conv1dSubsampling2 (k3, s2, p1)
            Tp = (T - 1) // 2 + 1
        else:
            Tp = ((T - 1) // 2 - 1) // 2
        enc_lens = np.zeros((B, ), dtype=np.int32)
        out = None
        if want_out:
            out = torch.empty((B, Tp, self._cfg.d_model), dtype=torch.float32,
                              device=self.device)
        _lib.check(
            self._L.wn_encode(self._h, speech.data_ptr(), _lib.i32p(lens), B,
                              T, chunk, left,
                              out.data_ptr() if out is not None else None,
                              _lib.i32p(enc_lens), _stream_ptr(self.device)),
            'wn_encode')
        return out, enc_lens, Tp

    def _forward_encoder(self, speech, speech_lengths,
                         decoding_chunk_size: int = -1,
                         num_decoding_left_chunks: int = -1,
                         simulate_streaming: bool = False):
        """asr_model.py:216-239 -> (encoder_out (B,T',d), encoder_mask (B,1,T'))."""
        if simulate_streaming and decoding_chunk_size > 0:
            self._check_simulate_streaming(speech)
        speech, lens = self._prep(speech, speech_lengths)
        out, enc_lens, Tp = self._encode(speech, lens, decoding_chunk_size,
                                         num_decoding_left_chunks, True)
        mask = (torch.arange(Tp, device=self.device).unsqueeze(0) <
                torch.as_tensor(enc_lens, device=self.device).unsqueeze(1))
        return out, mask.unsqueeze(1)

    # ---- streaming API: one chunk at a time with caches ----------------------
    def forward_encoder_chunk(self, xs: torch.Tensor, offset: int,
                              required_cache_size: int,
                              att_cache: Optional[torch.Tensor] = None,
                              cnn_cache: Optional[torch.Tensor] = None
                              ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """asr_model.py:385-427 / BaseEncoder.forward_chunk (encoder.py:204-285):
        xs (1, time, mel) on the device; att_cache (elayers, head, cache_t1,
        d_k * 2) and cnn_cache (elayers, 1, hidden, lorder), or empty / None for
        the first chunk -> (ys (1, chunk, hidden), new_att_cache, new_cnn_cache)
        in the reference's cache layouts (device tensors)."""
        cfg = self._cfg
        if cfg.encoder_type != 0:
            raise NotImplementedError('forward_encoder_chunk: Conformer encoders only')
        _require_cuda(xs, 'forward_encoder_chunk')
        assert xs.dim() == 3 and xs.size(0) == 1
        xs = xs.detach().to(torch.float32).contiguous()
        time = xs.size(1)
        chunk = ((time - 1) // 2 - 1) // 2
        L, H, d = cfg.n_layers, cfg.n_heads, cfg.d_model
        dk2 = 2 * d // H       # att_cache holds K | V per head (attention.py:226-234)
        lorder = cfg.cnn_kernel - 1 if cfg.causal else 0
        t1 = 0
        att_ptr = None
        if att_cache is not None and att_cache.numel() > 0:
            _require_cuda(att_cache, 'forward_encoder_chunk(att_cache)')
            assert tuple(att_cache.shape[:2]) == (L, H) and att_cache.size(3) == dk2
            att_cache = att_cache.detach().to(torch.float32).contiguous()
            t1 = att_cache.size(2)
            att_ptr = att_cache.data_ptr()
        cnn_ptr = None
        if cnn_cache is not None and cnn_cache.numel() > 0:
            _require_cuda(cnn_cache, 'forward_encoder_chunk(cnn_cache)')
            assert tuple(cnn_cache.shape) == (L, 1, d, lorder)
            cnn_cache = cnn_cache.detach().to(torch.float32).contiguous()
            cnn_ptr = cnn_cache.data_ptr()
        key = t1 + chunk
        if required_cache_size < 0:
            start = 0
        elif required_cache_size == 0:
            start = key
        else:
            start = max(key - required_cache_size, 0)
        ys = torch.empty((1, chunk, d), dtype=torch.float32, device=self.device)
        new_att = torch.empty((L, H, key - start, dk2), dtype=torch.float32,
                              device=self.device)
        if lorder > 0:
            new_cnn = torch.empty((L, 1, d, lorder), dtype=torch.float32,
                                  device=self.device)
        else:
            new_cnn = torch.zeros((L, 0, 0, 0), dtype=torch.float32, device=self.device)
        _lib.check(
            self._L.wn_encode_chunk(self._h, xs.data_ptr(), time, int(offset),
                                    int(required_cache_size), att_ptr, t1, cnn_ptr,
                                    ys.data_ptr(),
                                    new_att.data_ptr() if new_att.numel() else None,
                                    new_cnn.data_ptr() if lorder > 0 else None,
                                    None, None, _stream_ptr(self.device)),
            'wn_encode_chunk')
        return ys, new_att, new_cnn

    def forward_encoder_chunk_batch(self, xs: torch.Tensor, offsets, required_cache_size: int,
                                    att_caches=None, cnn_caches=None):
        """B streaming sessions in ONE forward_chunk (the reference's
        batched formulation is wenet/bin/export_onnx_gpu.py:83-232).  xs (B, time, mel):
        every session's window of this step (same length); offsets[b]: encoder frames
        session b has emitted; att_caches[b] / cnn_caches[b]: that session's caches in
        the reference's single-session layouts ((elayers, head, cache_t1_b, d_k * 2) --
        lengths may differ between sessions -- and (elayers, 1, hidden, lorder)), None /
        empty for a first chunk.  Returns (ys (B, chunk, hidden), [new_att_cache_b],
        [new_cnn_cache_b]); row b equals forward_encoder_chunk on session b alone."""
        import ctypes
        cfg = self._cfg
        if cfg.encoder_type != 0:
            raise NotImplementedError('forward_encoder_chunk: Conformer encoders only')
        _require_cuda(xs, 'forward_encoder_chunk_batch')
        assert xs.dim() == 3 and xs.size(0) == len(offsets)
        B = xs.size(0)
        xs = xs.detach().to(torch.float32).contiguous()
        time = xs.size(1)
        chunk = ((time - 1) // 2 - 1) // 2
        L, H, d = cfg.n_layers, cfg.n_heads, cfg.d_model
        dk2 = 2 * d // H
        lorder = cfg.cnn_kernel - 1 if cfg.causal else 0
        att_caches = list(att_caches) if att_caches is not None else [None] * B
        cnn_caches = list(cnn_caches) if cnn_caches is not None else [None] * B
        keep, t1s, att_ptrs, cnn_ptrs = [], [], [], []
        for b in range(B):
            a, cc = att_caches[b], cnn_caches[b]
            if a is not None and a.numel() > 0:
                _require_cuda(a, 'forward_encoder_chunk_batch(att_cache)')
                assert tuple(a.shape[:2]) == (L, H) and a.size(3) == dk2
                a = a.detach().to(torch.float32).contiguous()
                keep.append(a)
                t1s.append(a.size(2)); att_ptrs.append(a.data_ptr())
            else:
                t1s.append(0); att_ptrs.append(None)
            if cc is not None and cc.numel() > 0:
                _require_cuda(cc, 'forward_encoder_chunk_batch(cnn_cache)')
                assert tuple(cc.shape) == (L, 1, d, lorder)
                cc = cc.detach().to(torch.float32).contiguous()
                keep.append(cc)
                cnn_ptrs.append(cc.data_ptr())
            else:
                cnn_ptrs.append(None)
        new_att, new_cnn = [], []
        for b in range(B):
            key = t1s[b] + chunk
            if required_cache_size < 0:
                start = 0
            elif required_cache_size == 0:
                start = key
            else:
                start = max(key - required_cache_size, 0)
            new_att.append(torch.empty((L, H, key - start, dk2), dtype=torch.float32,
                                       device=self.device))
            new_cnn.append(torch.empty((L, 1, d, lorder), dtype=torch.float32,
                                       device=self.device) if lorder > 0 else
                           torch.zeros((L, 0, 0, 0), dtype=torch.float32, device=self.device))
        ys = torch.empty((B, chunk, d), dtype=torch.float32, device=self.device)
        vp = ctypes.c_void_p
        arr = lambda ptrs: (vp * B)(*[vp(p) if p else vp(None) for p in ptrs])  # noqa: E731
        offs = np.asarray(list(offsets), dtype=np.int32)
        t1a = np.asarray(t1s, dtype=np.int32)
        _lib.check(
            self._L.wn_encode_chunk_batch(
                self._h, B, xs.data_ptr(), time, _lib.i32p(offs), int(required_cache_size),
                arr(att_ptrs), _lib.i32p(t1a), arr(cnn_ptrs), ys.data_ptr(),
                arr([t.data_ptr() if t.numel() else None for t in new_att]),
                arr([t.data_ptr() if lorder > 0 else None for t in new_cnn]),
                None, None, _stream_ptr(self.device)), 'wn_encode_chunk_batch')
        del keep
        return ys, new_att, new_cnn

    def forward_encoder_chunk_by_chunk(self, xs: torch.Tensor, decoding_chunk_size: int,
                                       num_decoding_left_chunks: int = -1
                                       ) -> Tuple[torch.Tensor, torch.Tensor]:
        """BaseEncoder.forward_chunk_by_chunk (encoder.py:287-362) on the
        streaming API above: overlapping feature windows, caches carried from
        chunk to chunk -> (ys (1, T', hidden), masks (1, 1, T'))."""
        assert decoding_chunk_size > 0
        assert self._cfg.static_chunk_size > 0 or self._cfg.use_dynamic_chunk
        subsampling = self.subsampling_rate()
        context = self.right_context() + 1
        stride = subsampling * decoding_chunk_size
        window = (decoding_chunk_size - 1) * subsampling + context
        n = xs.size(1)
        att_cache = cnn_cache = None
        outs, offset = [], 0
        required = decoding_chunk_size * num_decoding_left_chunks
        for cur in range(0, n - context + 1, stride):
            y, att_cache, cnn_cache = self.forward_encoder_chunk(
                xs[:, cur:min(cur + window, n)], offset, required, att_cache, cnn_cache)
            outs.append(y)
            offset += y.size(1)
        ys = torch.cat(outs, 1)
        return ys, torch.ones((1, 1, ys.size(1)), dtype=torch.bool, device=ys.device)

    def ctc_activation(self, xs: torch.Tensor) -> torch.Tensor:
        """asr_model.py:429-440: CTC log-softmax of an encoder output."""
        return self.ctc_logprobs(xs)

    def _check_simulate_streaming(self, speech):
        """`simulate_streaming=True` (asr_model.py:229-233 ->
        BaseEncoder.forward_chunk_by_chunk, encoder.py:287-362) feeds the
        utterance chunk by chunk through attention / conv caches.  For a causal
        chunk-trained Conformer that is the SAME function as one pass under the
        chunk mask (every chunk sees exactly the cached left context the mask
        admits; the overlapping subsampling windows reproduce the full-utterance
        frames), which is what the accelerated encoder runs; the equivalence is
        pinned against the reference's own cache path
        (tests/golden/stream_*.npz).  Same preconditions as the reference."""
        assert speech.shape[0] == 1, 'forward_chunk_by_chunk: batch size must be 1'
        assert self._cfg.use_dynamic_chunk or self._cfg.static_chunk_size > 0, \
            'the model was not trained with static or dynamic chunks'
        if self._cfg.encoder_type != 0 or not self._cfg.causal:
            raise NotImplementedError(
                'simulate_streaming is accelerated for causal Conformer encoders '
                'only (a symmetric conv module sees no right context in '
                'forward_chunk)')

    def _set_encoder_out(self, encoder_out: torch.Tensor, encoder_lens):
        _require_cuda(encoder_out, '_set_encoder_out')
        enc = encoder_out.detach().to(torch.float32).contiguous()
        lens = torch.as_tensor(encoder_lens).detach().cpu().numpy().astype(
            np.int32)
        B, Tp, d = enc.shape
        assert d == self._cfg.d_model
        _lib.check(
            self._L.wn_set_encoder_out(self._h, enc.data_ptr(),
                                       _lib.i32p(lens), B, Tp,
                                       _stream_ptr(self.device)),
            'wn_set_encoder_out')
        return B, Tp

    def ctc_logprobs(self, encoder_out: torch.Tensor,
                     blank_penalty: float = 0.0, blank_id: int = 0,
                     encoder_lens=None):
        """asr_model.py:254-265 -> (B, T', V) log-probs in HBM."""
        B, Tp, _ = encoder_out.shape
        if encoder_lens is None:
            encoder_lens = torch.full((B, ), Tp, dtype=torch.int32)
        self._set_encoder_out(encoder_out, encoder_lens)
        out = torch.empty((B, Tp, self.vocab_size), dtype=torch.float32,
                          device=self.device)
        _lib.check(
            self._L.wn_ctc_logprobs(self._h, 1, blank_id, blank_penalty,
                                    out.data_ptr(), Tp,
                                    _stream_ptr(self.device)),
            'wn_ctc_logprobs')
        return out

    def _decoder_forward(self, which: int, tokens: torch.Tensor, lens: torch.Tensor
                         ) -> torch.Tensor:
        tok = np.ascontiguousarray(tokens.detach().cpu().numpy().astype(np.int32))
        ln = np.ascontiguousarray(torch.as_tensor(lens).detach().cpu().numpy()
                                  .astype(np.int32))
        n, L = tok.shape
        out = torch.empty((n, L, self.vocab_size), dtype=torch.float32,
                          device=self.device)
        _lib.check(
            self._L.wn_decoder_forward(self._h, 0, which, n, _lib.i32p(tok),
                                       _lib.i32p(ln), L, out.data_ptr(),
                                       _stream_ptr(self.device)),
            'wn_decoder_forward')
        return out

    def forward_attention_decoder(self, hyps: torch.Tensor, hyps_lens: torch.Tensor,
                                  encoder_out: torch.Tensor,
                                  reverse_weight: float = 0
                                  ) -> Tuple[torch.Tensor, torch.Tensor]:
        """asr_model.py:453-547: hyps (N, L) decoder inputs that already start
        with sos (eos-padded), hyps_lens (N,), encoder_out (1, T', d) ->
        (decoder_out, r_decoder_out), log-softmax over the vocabulary,
        (N, L, V) on the device; r_decoder_out is the scalar 0 the reference
        returns when the right-to-left decoder is not used."""
        assert encoder_out.size(0) == 1
        if self._cfg.dec_layers <= 0:
            raise RuntimeError('the model has no attention decoder on the accelerated path')
        self._set_encoder_out(encoder_out, [encoder_out.size(1)])
        hyps_c = hyps.detach().cpu().long()
        lens_c = torch.as_tensor(hyps_lens).detach().cpu().long()
        decoder_out = self._decoder_forward(0, hyps_c, lens_c)
        if reverse_weight > 0 and self.is_bidirectional_decoder():
            r_hyps = reverse_hyps(hyps_c, lens_c, self.eos)
            r_decoder_out = self._decoder_forward(1, r_hyps, lens_c)
        else:
            r_decoder_out = torch.tensor(0.0)
        return decoder_out, r_decoder_out

    def _rescore(self, ctc_prefix_results: List[DecodeResult],
                 ctc_weight: float, reverse_weight: float, raw=None):
        """attention_rescoring over the current batch.  `raw`: the arrays of the prefix beam
        search decode() just ran on this handle -- its n-best is still in HBM and is rescored
        there (no Python lists, no upload)."""
        if raw is not None:
            def times_of(b, i):
                return raw['hyp_times'][b, i, :raw['hyp_tlens'][b, i]].tolist()
            return rescore_nbest(self, raw['n_hyps'], raw['hyp_lens'], raw['hyp_tokens'],
                                 raw['hyp_scores'], times_of, ctc_weight, reverse_weight,
                                 True)
        n_hyps, hyp_lens, hyp_tokens, hyp_scores, times_of = _nbest_arrays(
            ctc_prefix_results)
        return rescore_nbest(self, n_hyps, hyp_lens, hyp_tokens, hyp_scores, times_of,
                             ctc_weight, reverse_weight, False)

    # ---- the drop-in entry point -----------------------------------------
    def decode(self,
               methods: List[str],
               speech: torch.Tensor,
               speech_lengths: torch.Tensor,
               beam_size: int = 1,
               decoding_chunk_size: int = -1,
               num_decoding_left_chunks: int = -1,
               ctc_weight: float = 0.0,
               simulate_streaming: bool = False,
               reverse_weight: float = 0.0,
               context_graph=None,
               blank_id: int = 0,
               blank_penalty: float = 0.0,
               length_penalty: float = 0.0,
               infos: Dict[str, List[str]] = None
               ) -> Dict[str, List[DecodeResult]]:
        """asr_model.py:267-343."""
        st = self._decode_begin(methods, speech, speech_lengths, beam_size,
                                decoding_chunk_size, num_decoding_left_chunks,
                                simulate_streaming, context_graph, blank_id,
                                blank_penalty)
        return self._decode_end(st, ctc_weight, reverse_weight, length_penalty)

    def _decode_begin(self, methods, speech, speech_lengths, beam_size=1,
                      decoding_chunk_size=-1, num_decoding_left_chunks=-1,
                      simulate_streaming=False, context_graph=None, blank_id=0,
                      blank_penalty=0.0):
        """First half of decode(): argument checks, then the encoder and the
        CTC head are queued on the current stream (no host sync)."""
        assert speech.shape[0] == speech_lengths.shape[0]
        assert decoding_chunk_size != 0
        self._use_context_graph(context_graph)
        if simulate_streaming and decoding_chunk_size > 0:
            self._check_simulate_streaming(speech)
        speech, lens = self._prep(speech, speech_lengths)
        B = speech.shape[0]
        _, enc_lens, Tp = self._encode(speech, lens, decoding_chunk_size,
                                       num_decoding_left_chunks, False)
        need_beam = ('ctc_prefix_beam_search' in methods
                     or 'attention_rescoring' in methods)
        k = beam_size if need_beam else 1
        _lib.check(
            self._L.wn_ctc_logprobs(self._h, k, blank_id, blank_penalty, None,
                                    Tp, _stream_ptr(self.device)),
            'wn_ctc_logprobs')
        return dict(methods=methods, B=B, enc_lens=enc_lens, need_beam=need_beam,
                    beam_size=beam_size, blank_id=blank_id, speech=speech, Tp=Tp)

    def _use_context_graph(self, graph):
        """Install / clear the biasing graph of the next prefix beam search
        (search.py:127-131 takes it as an argument; the C ABI keeps it on the
        handle).  The upload is skipped while the same graph object stays."""
        if graph is getattr(self, '_ctx_graph', None):
            return
        from wenet_amd import context_graph as cg
        cg.install(self._L, self._h, graph, _stream_ptr(self.device))
        self._ctx_graph = graph

    def _decode_end(self, st, ctc_weight=0.0, reverse_weight=0.0,
                    length_penalty=0.0):
        """Second half of decode(): the searches (+ rescoring) and the result
        records; synchronises the stream."""
        methods, B, enc_lens = st['methods'], st['B'], st['enc_lens']
        beam_size, blank_id = st['beam_size'], st['blank_id']
        results = {}
        self.last_non_blank_filter_empty = False
        max_len = int(enc_lens.max()) if B > 0 else 0
        if 'attention' in methods:
            if self._cfg.dec_layers <= 0:
                raise RuntimeError("'attention' mode: the model has no attention "
                                   'decoder on the accelerated path')
            if int(enc_lens.min()) <= 0:
                raise RuntimeError("'attention' mode: an utterance has no encoder "
                                   'frames')
            from wenet_amd.search import attention_beam_search
            # maxlen = encoder_out.size(1) of the padded reference tensor
            results['attention'] = attention_beam_search(
                self, B, st['Tp'], beam_size, length_penalty)
        if 'ctc_greedy_search' in methods:
            results['ctc_greedy_search'] = _greedy(self._h, B, max_len,
                                                   blank_id, self.device)
        prefix = None
        if st['need_beam']:
            if ('attention_rescoring' in methods and self._cfg.dec_layers > 0
                    and not self.apply_non_blank_embedding):
                # the hypothesis-independent part of rescoring runs on a second stream under the
                # prefix beam search (with the non-blank filter the decoder's memory is not
                # known yet)
                _lib.check(self._L.wn_rescore_prefetch(
                    self._h, 1 if reverse_weight > 0 else 0, _stream_ptr(self.device)),
                    'wn_rescore_prefetch')
            prefix, self._last_prefix_raw = _prefix_beam(
                self._h, B, max_len, beam_size, blank_id, self.device)
            if 'ctc_prefix_beam_search' in methods:
                results['ctc_prefix_beam_search'] = prefix
        if 'attention_rescoring' in methods:
            if self.apply_non_blank_embedding:
                # asr_model.py:337-342: the decoder's memory becomes the frames whose CTC
                # arg-max is not blank (+ the zero padding the reference leaves in)
                _, t_sel = self._filter_blank_current(B)
                # a batch without a single non-blank frame: the reference raises in
                # filter_blank_embedding; here the batch is rescored against the UNFILTERED
                # encoder output (deviations) and the caller can see that it was
                self.last_non_blank_filter_empty = t_sel == 0
                if t_sel == 0:
                    import warnings
                    warnings.warn('apply_non_blank_embedding: no non-blank frame in the whole '
                                  'batch; rescored against the unfiltered encoder output',
                                  RuntimeWarning, stacklevel=2)
            results['attention_rescoring'] = self._rescore(
                prefix, ctc_weight, reverse_weight, raw=self._last_prefix_raw)
        return results

    def _filter_blank_current(self, B: int, out: Optional[torch.Tensor] = None):
        """wn_filter_blank_embedding on the handle's current batch -> (kept rows per
        utterance, T)."""
        import ctypes
        n_keep = np.zeros((B, ), dtype=np.int32)
        t_out = ctypes.c_int32(0)
        _lib.check(
            self._L.wn_filter_blank_embedding(
                self._h, out.data_ptr() if out is not None else None, _lib.i32p(n_keep),
                ctypes.byref(t_out), _stream_ptr(self.device)), 'wn_filter_blank_embedding')
        return n_keep, int(t_out.value)

    def filter_blank_embedding(self, ctc_probs: torch.Tensor, encoder_out: torch.Tensor
                               ) -> Tuple[torch.Tensor, torch.Tensor]:
        """asr_model.py:153-180: (B, T, V) CTC posteriors + (B, T, d) encoder output ->
        the rows whose arg-max token is not 0, zero-padded to the longest selection, and the
        (B, 1, T_sel) mask of the kept rows.  Like the reference, every one of the T frames
        counts (the tensors carry no lengths)."""
        _require_cuda(encoder_out, 'filter_blank_embedding')
        _require_cuda(ctc_probs, 'filter_blank_embedding')
        B, T, _ = encoder_out.shape
        assert ctc_probs.shape[0] == B and ctc_probs.shape[1] == T
        lens = torch.full((B, ), T, dtype=torch.int32)
        self._set_encoder_out(encoder_out, lens)
        probs = ctc_probs.detach().to(torch.float32).contiguous()
        _lib.check(
            self._L.wn_set_ctc_probs(self._h, probs.data_ptr(), _lib.i32p(lens.numpy()), B, T,
                                     probs.shape[2], 1, _stream_ptr(self.device)),
            'wn_set_ctc_probs')
        out = torch.empty((B, T, encoder_out.shape[2]), dtype=torch.float32, device=self.device)
        n_keep, t_sel = self._filter_blank_current(B, out)
        if t_sel == 0:
            # every frame of every utterance is blank: nothing selected.  (The reference fails
            # here -- index_select with an empty float index; an empty selection with an
            # all-False mask is the value its code would have produced.)
            return (torch.zeros((B, 0, encoder_out.shape[2]), dtype=torch.float32,
                                device=self.device),
                    torch.zeros((B, 1, 0), dtype=torch.bool, device=self.device))
        sel = out.view(-1)[:B * t_sel * encoder_out.shape[2]].view(B, t_sel, -1)
        mask = (torch.arange(t_sel).unsqueeze(0) < torch.from_numpy(n_keep).unsqueeze(1))
        return sel, mask.unsqueeze(1).to(self.device)

    # older WeNet releases exposed decode() as recognize()
    def recognize(self, *args, **kwargs):
        return self.decode(*args, **kwargs)

    # ---- features + single-file CLI path -----------------------------------
    def _stage_pcm(self, waveforms, total: int) -> torch.Tensor:
        """Concatenated waveforms -> device through a ring of PINNED host buffers and an
        asynchronous copy on the current stream (a pageable `tensor.to(device)` stages through
        the driver's own bounce buffer and blocks the feeding thread for the whole transfer:
        20 MB per 32 x 10 s batch)."""
        ring = getattr(self, '_pcm_ring', None)
        if ring is None:
            ring = self._pcm_ring = dict(slot=0, bufs=[None] * 4, events=[None] * 4)
        i = ring['slot']
        ring['slot'] = (i + 1) % len(ring['bufs'])
        buf = ring['bufs'][i]
        if buf is None or buf.numel() < total:
            buf = ring['bufs'][i] = torch.empty(max(total, 1) * 5 // 4 + 1024,
                                                 dtype=torch.float32).pin_memory()
        elif ring['events'][i] is not None:
            ring['events'][i].synchronize()       # the copy that last used this buffer is done
        view = buf.numpy()
        pool = getattr(self, 'stage_pool', None)
        if pool is not None and len(waveforms) >= 8:
            # the bulk copies of a batch (20 MB for 32 x 10 s) on the caller's thread pool: numpy
            # drops the GIL in them (wenet_amd/bin/recognize.py sets `stage_pool`)
            offs = np.zeros((len(waveforms) + 1, ), dtype=np.int64)
            offs[1:] = np.cumsum([len(w) for w in waveforms])

            def put(lo, hi):
                for i in range(lo, hi):
                    view[offs[i]:offs[i + 1]] = waveforms[i]
            step = (len(waveforms) + 3) // 4
            futs = [pool.submit(put, lo, min(lo + step, len(waveforms)))
                    for lo in range(0, len(waveforms), step)]
            for f in futs:
                f.result()
        else:
            pos = 0
            for w in waveforms:
                n = len(w)
                view[pos:pos + n] = w
                pos += n
        with torch.cuda.device(self.device):
            pcm = buf[:total].to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
        ring['events'][i] = ev
        return pcm

    def compute_fbank(self, waveforms: List[np.ndarray]
                      ) -> Tuple[torch.Tensor, torch.Tensor]:
        """processor.compute_fbank + padding for a list of float waveforms in
        [-1, 1] (16 kHz) -> ((B, Tmax, F) features in HBM, lengths)."""
        B = len(waveforms)
        offs = np.zeros((B + 1, ), dtype=np.int64)
        for i, w in enumerate(waveforms):
            offs[i + 1] = offs[i] + len(w)
        pcm = self._stage_pcm(waveforms, int(offs[B]))
        nfr = [0 if len(w) < 400 else 1 + (len(w) - 400) // 160
               for w in waveforms]
        tmax = max(max(nfr), 1)
        feats = torch.empty((B, tmax, self._cfg.feat_dim), dtype=torch.float32,
                            device=self.device)
        n_frames = np.zeros((B, ), dtype=np.int32)
        _lib.check(
            self._L.wn_fbank(self._h, pcm.data_ptr(), _lib.i64p(offs), B,
                             feats.data_ptr(), tmax, _lib.i32p(n_frames),
                             _stream_ptr(self.device)), 'wn_fbank')
        return feats, torch.from_numpy(n_frames.copy())

    def compute_log_mel_spectrogram(self, waveforms: List[np.ndarray],
                                    num_mel_bins: int = 80, padding: int = 0,
                                    pad_or_trim: bool = False,
                                    max_duration: int = 30
                                    ) -> Tuple[torch.Tensor, torch.Tensor]:
        """processor.compute_log_mel_spectrogram (processor.py:320-369, the
        Whisper frontend) for a list of float waveforms in [-1, 1] (16 kHz)
        -> ((B, Tmax, num_mel_bins) features in HBM, lengths)."""
        ws = []
        for w in waveforms:
            w = np.asarray(w, np.float32)
            if padding > 0:
                w = np.concatenate([w, np.zeros(padding, np.float32)])
            if pad_or_trim:
                n = max_duration * 16000
                w = w[:n] if len(w) >= n else np.concatenate(
                    [w, np.zeros(n - len(w), np.float32)])
            ws.append(w)
        B = len(ws)
        offs = np.zeros((B + 1, ), dtype=np.int64)
        offs[1:] = np.cumsum([len(w) for w in ws])
        pcm = torch.from_numpy(np.concatenate(ws)).to(self.device)
        tmax = max(max(len(w) // 160 for w in ws), 1)
        feats = torch.empty((B, tmax, num_mel_bins), dtype=torch.float32,
                            device=self.device)
        n_frames = np.zeros((B, ), dtype=np.int32)
        _lib.check(
            self._L.wn_log_mel(self._h, pcm.data_ptr(), _lib.i64p(offs), B,
                               num_mel_bins, feats.data_ptr(), tmax,
                               _lib.i32p(n_frames), _stream_ptr(self.device)),
            'wn_log_mel')
        return feats, torch.from_numpy(n_frames.copy())

    def resample(self, waveform, orig_freq: int, new_freq: int = 16000) -> np.ndarray:
        """processor.resample (processor.py:177-196; torchaudio Resample
        defaults) of one float waveform on the device -> host float32 array."""
        x = torch.as_tensor(np.asarray(waveform, dtype=np.float32)).to(self.device)
        n_in = int(x.numel())
        n_out = int(self._L.wn_resample_length(n_in, int(orig_freq), int(new_freq)))
        out = torch.empty((n_out, ), dtype=torch.float32, device=self.device)
        if n_out > 0:
            _lib.check(
                self._L.wn_resample(self._h, x.data_ptr(), n_in, int(orig_freq),
                                    int(new_freq), out.data_ptr(), n_out,
                                    _stream_ptr(self.device)), 'wn_resample')
        return out.cpu().numpy()

    def load_wav(self, wav_file: str) -> np.ndarray:
        """decode_wav + resample to 16 kHz (processor.py:125-153,177-196)."""
        wav, rate = read_wav(wav_file, return_rate=True)
        return wav if rate == 16000 else self.resample(wav, rate, 16000)

    def compute_feature(self, wav_file: str) -> torch.Tensor:
        """cli/model.py:47-66 (load_feature): decode_wav -> resample -> the
        feature function `dataset_conf.feats_type` names with its `<type>_conf`
        -- `fbank` (default) or `log_mel_spectrogram` (the Whisper recipes);
        `mfcc` is not on this path."""
        dc = self.configs.get('dataset_conf') or {}
        feats_type = dc.get('feats_type', 'fbank')
        if feats_type == 'fbank':
            feats, lens = self.compute_fbank([self.load_wav(wav_file)])
        elif feats_type == 'log_mel_spectrogram':
            conf = dc.get('log_mel_spectrogram_conf') or {}
            if conf.get('n_fft', 400) != 400 or conf.get('hop_length', 160) != 160:
                raise NotImplementedError(
                    'log_mel_spectrogram: n_fft 400 / hop_length 160 only')
            feats, lens = self.compute_log_mel_spectrogram(
                [self.load_wav(wav_file)],
                num_mel_bins=conf.get('num_mel_bins', 80),
                padding=conf.get('padding', 0),
                pad_or_trim=conf.get('pad_or_trim', False),
                max_duration=conf.get('max_duration', 30))
        else:
            raise NotImplementedError(
                f'feats_type {feats_type!r}: fbank and log_mel_spectrogram only')
        return feats[0, :int(lens[0])]

    def transcribe(self, wav: str) -> DecodeResult:
        """asr_model.py:345-358 (decode defaults: beam_size=1, ctc_weight=0)."""
        assert hasattr(self, 'tokenizer')
        speech = self.compute_feature(wav)
        speech_lengths = torch.tensor([speech.size(0)])
        results = self.decode([self.default_decode_method],
                              speech.unsqueeze(0), speech_lengths)
        result = results[self.default_decode_method][0]
        result.text = self.tokenizer.detokenize(result.tokens)[0]
        return result


def _parse_riff_wave(buf: bytes):
    """RIFF/WAVE container -> (format tag, channels, rate, bits, data bytes).
    PCM (1), IEEE float (3) and WAVE_FORMAT_EXTENSIBLE (0xFFFE, sub-format in
    the GUID's first two bytes); unknown chunks (LIST, fact, ...) are skipped."""
    import struct
    if len(buf) < 12 or buf[:4] != b'RIFF' or buf[8:12] != b'WAVE':
        raise ValueError('not a RIFF/WAVE file')
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(buf):
        cid, size = buf[pos:pos + 4], struct.unpack('<I', buf[pos + 4:pos + 8])[0]
        body = buf[pos + 8:pos + 8 + size]
        if cid == b'fmt ':
            tag, nch, rate, _, _, bits = struct.unpack('<HHIIHH', body[:16])
            if tag == 0xFFFE and len(body) >= 26:
                tag = struct.unpack('<H', body[24:26])[0]
            fmt = (tag, nch, rate, bits)
        elif cid == b'data':
            data = body  # a truncated file yields the samples that are there
            if fmt is not None:
                break
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError('wav file without fmt / data chunk')
    return fmt + (data, )


def read_wav(path, return_rate: bool = False, start: Optional[float] = None,
             end: Optional[float] = None):
    """wav file (path, bytes or file object) -> mono (first channel,
    processor.singal_channel) float32 in [-1, 1) with torchaudio.load's
    normalisation (processor.decode_wav, processor.py:125-153): u8 -> (x-128)/128,
    s16 / s24 / s32 -> x / 2^(bits-1), float32 / float64 as stored.  `start` /
    `end` (seconds) select a segment like the raw lists' `start` / `end` keys.
    Without `return_rate` the file must be 16 kHz (use ASRModel.load_wav to
    resample).  Compressed containers (flac, mp3, ...) are not decoded."""
    if isinstance(path, (bytes, bytearray, memoryview)):
        buf = bytes(path)
    elif hasattr(path, 'read'):
        buf = path.read()
    else:
        with open(path, 'rb') as f:
            buf = f.read()
    tag, nch, sr, bits, raw = _parse_riff_wave(buf)
    frame = nch * (bits // 8)
    raw = raw[:len(raw) - len(raw) % max(frame, 1)]
    if tag == 1 and bits == 16:
        data = np.frombuffer(raw, dtype='<i2').astype(np.float32) / 32768.0
    elif tag == 1 and bits == 8:
        data = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif tag == 1 and bits == 24:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        data = v.astype(np.float32) / float(1 << 23)
    elif tag == 1 and bits == 32:
        data = (np.frombuffer(raw, dtype='<i4').astype(np.float64) / 2147483648.0
                ).astype(np.float32)
    elif tag == 3 and bits == 32:
        data = np.frombuffer(raw, dtype='<f4').astype(np.float32)
    elif tag == 3 and bits == 64:
        data = np.frombuffer(raw, dtype='<f8').astype(np.float32)
    else:
        raise NotImplementedError(
            f'wav format tag {tag} with {bits} bits is not supported (PCM 8/16/24/32, '
            'IEEE float 32/64)')
    if nch > 1:
        data = np.ascontiguousarray(data.reshape(-1, nch)[:, 0])
    if start is not None:
        assert end is not None  # processor.py:139
        s0, s1 = int(start * sr), int(end * sr)
        data = data[s0:s1]
    if return_rate:
        return data, sr
    if sr != 16000:
        raise NotImplementedError('read_wav: not 16 kHz; use ASRModel.load_wav')
    return data


def load_model(model_name_or_path: str, device='cuda') -> ASRModel:
    """wenet/cli/model.py:71-110 for a local model directory holding
    train.yaml, final.pt, units.txt and (optionally) global_cmvn."""
    import yaml
    model_dir = model_name_or_path
    for f in ('train.yaml', 'final.pt', 'units.txt'):
        if not os.path.exists(os.path.join(model_dir, f)):
            raise FileNotFoundError(
                f'Required file {f} not found in {model_dir}')
    with open(os.path.join(model_dir, 'train.yaml'), 'r') as fin:
        configs = yaml.load(fin, Loader=yaml.FullLoader)
    sd = torch.load(os.path.join(model_dir, 'final.pt'), map_location='cpu',
                    mmap=True)
    sd = dict(sd)
    if configs.get('cmvn') == 'global_cmvn' and \
            'encoder.global_cmvn.mean' not in sd:
        cmvn_file = os.path.join(model_dir, 'global_cmvn')
        if not os.path.exists(cmvn_file):
            cmvn_file = configs['cmvn_conf']['cmvn_file']
        mean, istd = load_cmvn(cmvn_file, configs['cmvn_conf']['is_json_cmvn'])
        sd['encoder.global_cmvn.mean'] = torch.from_numpy(mean).float()
        sd['encoder.global_cmvn.istd'] = torch.from_numpy(istd).float()
    model = ASRModel(configs, sd, device)
    # cli/model.py:35-46: the config's tokenizer with its files looked up in the
    # model directory; older packages without tokenizer_conf fall back to units.txt
    if 'tokenizer_conf' in configs and \
            configs['tokenizer_conf'].get('symbol_table_path') is not None:
        from wenet_amd.tokenizer import init_tokenizer
        model.tokenizer = init_tokenizer(configs, model_dir)
    else:
        model.tokenizer = _Tokenizer(os.path.join(model_dir, 'units.txt'))
    return model
