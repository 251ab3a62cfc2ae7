"""Synthetic model directories, weights and inputs for tests and bench.py.

There is no network for checkpoints or datasets, so the named reference
configurations are rebuilt here from their yaml (the dicts below restate
/root/reference/examples/*/s0/conf/*.yaml, cited per entry) and filled with
deterministic random weights whose names and shapes are exactly the reference
``state_dict`` (tests/test_oracle_vs_reference.py checks that against the real
``init_model`` when the reference tree is present).

Weights are drawn per tensor from ``numpy.random.Generator(PCG64)`` keyed by
(seed, tensor name) so any process -- this container, the GPU box, a different
rank -- regenerates bit-identical tensors without shipping a 200 MB file.

The CTC head is "sharpened": a Xavier-random CTC projection gives near-uniform
posteriors whose per-frame argmax margin (~1e-5) is below fp32 reordering noise
(hard part 1) and whose greedy output is ~T' tokens long,
which no real model produces.  We scale the projection and bias blank so that
about 3/4 of frames are blank and the top-1 margin is O(1), like a trained CTC
model.
"""
import copy
import math
import os
import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np
import torch

_SPECIAL = {'<blank>': 0, '<unk>': 1, '<sos>': 2, '<eos>': 2}

CONFIGS = {
    # examples/aishell/s0/conf/train_u2++_conformer.yaml:4-62
    'aishell_u2pp': {
        'input_dim': 80, 'output_dim': 4233,
        'encoder': 'conformer',
        'encoder_conf': dict(output_size=256, attention_heads=4,
                             linear_units=2048, num_blocks=12,
                             dropout_rate=0.1, positional_dropout_rate=0.1,
                             attention_dropout_rate=0.1, input_layer='conv2d',
                             normalize_before=True, cnn_module_kernel=8,
                             use_cnn_module=True, activation_type='swish',
                             pos_enc_layer_type='rel_pos',
                             selfattention_layer_type='rel_selfattn',
                             causal=True, use_dynamic_chunk=True,
                             cnn_module_norm='layer_norm',
                             use_dynamic_left_chunk=False),
        'decoder': 'bitransformer',
        'decoder_conf': dict(attention_heads=4, linear_units=2048,
                             num_blocks=3, r_num_blocks=3, dropout_rate=0.1,
                             positional_dropout_rate=0.1,
                             self_attention_dropout_rate=0.1,
                             src_attention_dropout_rate=0.1),
        'model_conf': dict(ctc_weight=0.3, lsm_weight=0.1,
                           length_normalized_loss=False, reverse_weight=0.3),
    },
    # examples/aishell/s0/conf/train_u2++_lite_conformer.yaml:1-68 at its own widths
    'aishell_u2pp_lite': {
        'input_dim': 80, 'output_dim': 4233,
        'encoder': 'conformer',
        'encoder_conf': dict(output_size=256, attention_heads=4,
                             linear_units=2048, num_blocks=12,
                             dropout_rate=0.1, positional_dropout_rate=0.1,
                             attention_dropout_rate=0.1, input_layer='conv2d',
                             normalize_before=True, cnn_module_kernel=8,
                             use_cnn_module=True, activation_type='swish',
                             pos_enc_layer_type='rel_pos',
                             selfattention_layer_type='rel_selfattn',
                             causal=True, use_dynamic_chunk=True,
                             cnn_module_norm='layer_norm',
                             use_dynamic_left_chunk=False),
        'decoder': 'bitransformer',
        'decoder_conf': dict(attention_heads=4, linear_units=1024,
                             num_blocks=3, r_num_blocks=3, dropout_rate=0.1,
                             positional_dropout_rate=0.1,
                             self_attention_dropout_rate=0.1,
                             src_attention_dropout_rate=0.1),
        'model_conf': dict(ctc_weight=0.3, lsm_weight=0.1,
                           length_normalized_loss=False, reverse_weight=0.3,
                           apply_non_blank_embedding=True),
    },
    # the same recipe at tiny widths for tests
    'tiny_lite': {
        'input_dim': 80, 'output_dim': 53,
        'encoder': 'conformer',
        'encoder_conf': dict(output_size=64, attention_heads=1,
                             linear_units=128, num_blocks=2,
                             dropout_rate=0.1, positional_dropout_rate=0.1,
                             attention_dropout_rate=0.1, input_layer='conv2d',
                             normalize_before=True, cnn_module_kernel=8,
                             use_cnn_module=True, activation_type='swish',
                             pos_enc_layer_type='rel_pos',
                             selfattention_layer_type='rel_selfattn',
                             causal=True, use_dynamic_chunk=True,
                             cnn_module_norm='layer_norm',
                             use_dynamic_left_chunk=False),
        'decoder': 'bitransformer',
        'decoder_conf': dict(attention_heads=1, linear_units=96,
                             num_blocks=2, r_num_blocks=1, dropout_rate=0.1,
                             positional_dropout_rate=0.1,
                             self_attention_dropout_rate=0.1,
                             src_attention_dropout_rate=0.1),
        'model_conf': dict(ctc_weight=0.3, lsm_weight=0.1,
                           length_normalized_loss=False, reverse_weight=0.3,
                           apply_non_blank_embedding=True),
    },
    # examples/librispeech/s0/conf/train_conformer_bidecoder_large.yaml:3-62
    'librispeech_bidecoder_large': {
        'input_dim': 80, 'output_dim': 5002,
        'encoder': 'conformer',
        'encoder_conf': dict(output_size=512, attention_heads=8,
                             linear_units=2048, num_blocks=12,
                             dropout_rate=0.1, positional_dropout_rate=0.1,
                             attention_dropout_rate=0.1, input_layer='conv2d',
                             normalize_before=True, cnn_module_kernel=31,
                             use_cnn_module=True, activation_type='swish',
                             pos_enc_layer_type='rel_pos',
                             selfattention_layer_type='rel_selfattn',
                             cnn_module_norm='layer_norm'),
        'decoder': 'bitransformer',
        'decoder_conf': dict(attention_heads=8, linear_units=2048,
                             num_blocks=3, r_num_blocks=3, dropout_rate=0.1,
                             positional_dropout_rate=0.1,
                             self_attention_dropout_rate=0.1,
                             src_attention_dropout_rate=0.1),
        'model_conf': dict(ctc_weight=0.3, lsm_weight=0.1,
                           length_normalized_loss=False, reverse_weight=0.3),
    },
    # examples/wenetspeech/s0/conf/train_u2++_conformer.yaml:1-60
    'wenetspeech_u2pp': {
        'input_dim': 80, 'output_dim': 5538,
        'encoder': 'conformer',
        'encoder_conf': dict(activation_type='swish',
                             attention_dropout_rate=0.1, attention_heads=8,
                             causal=True, cnn_module_kernel=15,
                             cnn_module_norm='layer_norm', dropout_rate=0.1,
                             input_layer='conv2d', linear_units=2048,
                             normalize_before=True, num_blocks=12,
                             output_size=512, pos_enc_layer_type='rel_pos',
                             positional_dropout_rate=0.1,
                             selfattention_layer_type='rel_selfattn',
                             use_cnn_module=True, use_dynamic_chunk=True,
                             use_dynamic_left_chunk=False),
        'decoder': 'bitransformer',
        'decoder_conf': dict(attention_heads=8, dropout_rate=0.1,
                             linear_units=2048, num_blocks=3,
                             positional_dropout_rate=0.1, r_num_blocks=3,
                             self_attention_dropout_rate=0.1,
                             src_attention_dropout_rate=0.1),
        'model_conf': dict(ctc_weight=0.3, lsm_weight=0.1,
                           length_normalized_loss=False, reverse_weight=0.3),
    },
    # miniature of the u2++ recipe (causal conv, dynamic chunk, bi-decoder)
    'tiny_causal': {
        'input_dim': 80, 'output_dim': 67,
        'encoder': 'conformer',
        'encoder_conf': dict(output_size=128, attention_heads=2,
                             linear_units=192, num_blocks=2, dropout_rate=0.1,
                             positional_dropout_rate=0.1,
                             attention_dropout_rate=0.1, input_layer='conv2d',
                             normalize_before=True, cnn_module_kernel=8,
                             use_cnn_module=True, activation_type='swish',
                             pos_enc_layer_type='rel_pos',
                             selfattention_layer_type='rel_selfattn',
                             causal=True, use_dynamic_chunk=True,
                             cnn_module_norm='layer_norm',
                             use_dynamic_left_chunk=False),
        'decoder': 'bitransformer',
        'decoder_conf': dict(attention_heads=2, linear_units=160,
                             num_blocks=2, r_num_blocks=1, dropout_rate=0.1,
                             positional_dropout_rate=0.1,
                             self_attention_dropout_rate=0.1,
                             src_attention_dropout_rate=0.1),
        'model_conf': dict(ctc_weight=0.3, lsm_weight=0.1,
                           length_normalized_loss=False, reverse_weight=0.3),
    },
    # examples/aishell/whisper/conf/finetune_whisper_largev3.yaml:1-17 (encoder)
    # at 2 of its 32 blocks: the Whisper-large-v3 widths (1280 / 20 heads /
    # 5120, 128 mel bins) at a size a CPU reference run can afford.  Encoder +
    # CTC head only; the Whisper decoder is outside the accelerated path.
    'whisper_largev3_2blocks': {
        'input_dim': 128, 'output_dim': 307,
        'encoder': 'transformer',
        'encoder_conf': dict(activation_type='gelu', attention_dropout_rate=0.0,
                             attention_heads=20, dropout_rate=0.0,
                             input_layer='conv1d2', key_bias=False,
                             linear_units=5120, normalize_before=True,
                             num_blocks=2, output_size=1280,
                             pos_enc_layer_type='abs_pos_whisper',
                             positional_dropout_rate=0.0, static_chunk_size=-1,
                             use_dynamic_chunk=False,
                             use_dynamic_left_chunk=False),
        'decoder': None, 'decoder_conf': {},
        'cmvn': None,
        'model_conf': dict(ctc_weight=0.3, lsm_weight=0.1,
                           length_normalized_loss=False),
    },
    # the full Whisper-large-v3 encoder (32 blocks; 0.63 G parameters) for
    # bench.py --workload config5
    'whisper_largev3': {
        'input_dim': 128, 'output_dim': 307,
        'encoder': 'transformer',
        'encoder_conf': dict(activation_type='gelu', attention_dropout_rate=0.0,
                             attention_heads=20, dropout_rate=0.0,
                             input_layer='conv1d2', key_bias=False,
                             linear_units=5120, normalize_before=True,
                             num_blocks=32, output_size=1280,
                             pos_enc_layer_type='abs_pos_whisper',
                             positional_dropout_rate=0.0, static_chunk_size=-1,
                             use_dynamic_chunk=False,
                             use_dynamic_left_chunk=False),
        'decoder': None, 'decoder_conf': {},
        'cmvn': None,
        'model_conf': dict(ctc_weight=0.3, lsm_weight=0.1,
                           length_normalized_loss=False),
    },
    # miniature of the same encoder family (whisper-tiny widths: 384 / 6 heads)
    'whisper_tiny_like': {
        'input_dim': 80, 'output_dim': 211,
        'encoder': 'transformer',
        'encoder_conf': dict(activation_type='gelu', attention_dropout_rate=0.0,
                             attention_heads=6, dropout_rate=0.0,
                             input_layer='conv1d2', key_bias=False,
                             linear_units=1536, normalize_before=True,
                             num_blocks=4, output_size=384,
                             pos_enc_layer_type='abs_pos_whisper',
                             positional_dropout_rate=0.0, static_chunk_size=-1,
                             use_dynamic_chunk=False,
                             use_dynamic_left_chunk=False),
        'decoder': None, 'decoder_conf': {},
        'cmvn': None,
        'model_conf': dict(ctc_weight=0.3, lsm_weight=0.1,
                           length_normalized_loss=False),
    },
    # examples/aishell/s0/conf/train_conformer.yaml:1-35: the offline recipe --
    # symmetric conv (kernel 15), cnn_module_norm left at its default
    # 'batch_norm', left-to-right decoder
    'aishell_conformer': {
        'input_dim': 80, 'output_dim': 4233,
        'encoder': 'conformer',
        'encoder_conf': dict(output_size=256, attention_heads=4,
                             linear_units=2048, num_blocks=12, dropout_rate=0.1,
                             positional_dropout_rate=0.1,
                             attention_dropout_rate=0.0, input_layer='conv2d',
                             normalize_before=True, cnn_module_kernel=15,
                             use_cnn_module=True, activation_type='swish',
                             pos_enc_layer_type='rel_pos',
                             selfattention_layer_type='rel_selfattn'),
        'decoder': 'transformer',
        'decoder_conf': dict(attention_heads=4, linear_units=2048, num_blocks=6,
                             dropout_rate=0.1, positional_dropout_rate=0.1,
                             self_attention_dropout_rate=0.0,
                             src_attention_dropout_rate=0.0),
        'model_conf': dict(ctc_weight=0.3, lsm_weight=0.1,
                           length_normalized_loss=False),
    },
    # miniature of it (batch_norm in the conv module)
    'tiny_bn': {
        'input_dim': 80, 'output_dim': 89,
        'encoder': 'conformer',
        'encoder_conf': dict(output_size=128, attention_heads=2,
                             linear_units=160, num_blocks=2, dropout_rate=0.1,
                             positional_dropout_rate=0.1,
                             attention_dropout_rate=0.0, input_layer='conv2d',
                             normalize_before=True, cnn_module_kernel=15,
                             use_cnn_module=True, activation_type='swish',
                             pos_enc_layer_type='rel_pos',
                             selfattention_layer_type='rel_selfattn'),
        'decoder': 'transformer',
        'decoder_conf': dict(attention_heads=2, linear_units=128, num_blocks=2,
                             dropout_rate=0.1, positional_dropout_rate=0.1,
                             self_attention_dropout_rate=0.0,
                             src_attention_dropout_rate=0.0),
        'model_conf': dict(ctc_weight=0.3, lsm_weight=0.1,
                           length_normalized_loss=False),
    },
    # miniature of the offline recipe (symmetric conv, full attention,
    # left-to-right decoder only)
    'tiny_sym': {
        'input_dim': 80, 'output_dim': 101,
        'encoder': 'conformer',
        'encoder_conf': dict(output_size=64, attention_heads=1,
                             linear_units=128, num_blocks=3, dropout_rate=0.1,
                             positional_dropout_rate=0.1,
                             attention_dropout_rate=0.1, input_layer='conv2d',
                             normalize_before=True, cnn_module_kernel=15,
                             use_cnn_module=True, activation_type='swish',
                             pos_enc_layer_type='rel_pos',
                             selfattention_layer_type='rel_selfattn',
                             cnn_module_norm='layer_norm'),
        'decoder': 'transformer',
        'decoder_conf': dict(attention_heads=1, linear_units=96, num_blocks=2,
                             dropout_rate=0.1, positional_dropout_rate=0.1,
                             self_attention_dropout_rate=0.1,
                             src_attention_dropout_rate=0.1),
        'model_conf': dict(ctc_weight=0.3, lsm_weight=0.1,
                           length_normalized_loss=False),
    },
}


def make_configs(name: str) -> dict:
    """Parsed-``train.yaml`` equivalent for a named configuration."""
    c = copy.deepcopy(CONFIGS[name])
    c.setdefault('ctc', 'ctc')
    c.setdefault('ctc_conf', {'ctc_blank_id': 0})
    c.setdefault('model', 'asr_model')
    c.setdefault('cmvn', 'global_cmvn')
    c.setdefault('cmvn_conf', {'cmvn_file': 'global_cmvn',
                               'is_json_cmvn': True})
    c.setdefault('tokenizer', 'char')
    c.setdefault('tokenizer_conf', {'symbol_table_path': 'units.txt',
                                    'split_with_space': False,
                                    'bpe_path': None,
                                    'non_lang_syms_path': None,
                                    'is_multilingual': False,
                                    'num_languages': 1,
                                    'special_tokens': dict(_SPECIAL)})
    c.setdefault('dataset', 'asr')
    c.setdefault('dataset_conf', {
        'resample_conf': {'resample_rate': 16000},
        'fbank_conf': {'num_mel_bins': 80, 'frame_shift': 10,
                       'frame_length': 25, 'dither': 0.0},
    })
    return c


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(
        np.random.PCG64([seed, zlib.crc32(name.encode('utf-8'))]))


def _uniform(seed, name, shape, bound):
    return _rng(seed, name).uniform(-bound, bound, size=shape).astype(
        np.float32)


def _normal(seed, name, shape, std=1.0, mean=0.0):
    return (_rng(seed, name).standard_normal(size=shape) * std + mean).astype(
        np.float32)


def positional_table(d_model: int, max_len: int = 5000) -> torch.Tensor:
    """The `pe` buffer of wenet/models/transformer/embedding.py:47-56."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(
        torch.arange(0, d_model, 2, dtype=torch.float32) *
        -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0)


def make_state_dict(configs: dict, seed: int = 0,
                    sharpen_ctc: bool = True) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic random weights with the reference's names and shapes."""
    ec, dc = configs['encoder_conf'], configs['decoder_conf']
    d = ec['output_size']
    h = ec['attention_heads']
    ffn = ec['linear_units']
    K = ec.get('cnn_module_kernel', 15)
    if configs.get('encoder', 'conformer') == 'transformer':
        return _make_transformer_state_dict(configs, seed, sharpen_ctc)
    idim, V = configs['input_dim'], configs['output_dim']
    sd: Dict[str, np.ndarray] = OrderedDict()

    def lin(name, out_f, in_f, bias=True):
        b = 1.0 / math.sqrt(in_f)
        sd[name + '.weight'] = _uniform(seed, name + '.weight', (out_f, in_f), b)
        if bias:
            sd[name + '.bias'] = _uniform(seed, name + '.bias', (out_f, ), b)

    def norm(name, n):
        sd[name + '.weight'] = _normal(seed, name + '.weight', (n, ), 0.1, 1.0)
        sd[name + '.bias'] = _normal(seed, name + '.bias', (n, ), 0.1)

    def attn(name, n, rel=False):
        if rel:
            b = math.sqrt(6.0 / (h + n // h))
            sd[name + '.pos_bias_u'] = _uniform(seed, name + '.pos_bias_u',
                                                (h, n // h), b)
            sd[name + '.pos_bias_v'] = _uniform(seed, name + '.pos_bias_v',
                                                (h, n // h), b)
        for p in ('linear_q', 'linear_k', 'linear_v', 'linear_out'):
            lin(f'{name}.{p}', n, n)
        if rel:
            lin(f'{name}.linear_pos', n, n, bias=False)

    if configs.get('cmvn') == 'global_cmvn':
        sd['encoder.global_cmvn.mean'] = _normal(seed, 'cmvn.mean', (idim, ),
                                                 1.5, 11.0)
        sd['encoder.global_cmvn.istd'] = (
            1.0 / (2.5 + np.abs(_normal(seed, 'cmvn.std', (idim, ), 0.5)))
        ).astype(np.float32)
    b = 1.0 / 3.0
    sd['encoder.embed.conv.0.weight'] = _uniform(seed, 'conv0.w', (d, 1, 3, 3), b)
    sd['encoder.embed.conv.0.bias'] = _uniform(seed, 'conv0.b', (d, ), b)
    b = 1.0 / math.sqrt(9 * d)
    sd['encoder.embed.conv.2.weight'] = _uniform(seed, 'conv2.w', (d, d, 3, 3), b)
    sd['encoder.embed.conv.2.bias'] = _uniform(seed, 'conv2.b', (d, ), b)
    fdim = ((idim - 1) // 2 - 1) // 2
    lin('encoder.embed.out.0', d, d * fdim)
    sd['encoder.embed.pos_enc.pe'] = positional_table(d).numpy()
    norm('encoder.after_norm', d)
    for i in range(ec['num_blocks']):
        p = f'encoder.encoders.{i}'
        attn(p + '.self_attn', d, rel=True)
        for ff in ('feed_forward', 'feed_forward_macaron'):
            lin(f'{p}.{ff}.w_1', ffn, d)
            lin(f'{p}.{ff}.w_2', d, ffn)
        b = 1.0 / math.sqrt(d)
        sd[p + '.conv_module.pointwise_conv1.weight'] = _uniform(
            seed, p + '.pw1.w', (2 * d, d, 1), b)
        sd[p + '.conv_module.pointwise_conv1.bias'] = _uniform(
            seed, p + '.pw1.b', (2 * d, ), b)
        b = 1.0 / math.sqrt(K)
        sd[p + '.conv_module.depthwise_conv.weight'] = _uniform(
            seed, p + '.dw.w', (d, 1, K), b)
        sd[p + '.conv_module.depthwise_conv.bias'] = _uniform(
            seed, p + '.dw.b', (d, ), b)
        norm(p + '.conv_module.norm', d)
        if ec.get('cnn_module_norm', 'batch_norm') == 'batch_norm':
            # BatchNorm1d buffers (eval mode uses the running statistics)
            sd[p + '.conv_module.norm.running_mean'] = _normal(
                seed, p + '.bn.mean', (d, ), 0.3)
            sd[p + '.conv_module.norm.running_var'] = (
                0.5 + np.abs(_normal(seed, p + '.bn.var', (d, ), 0.5))).astype(np.float32)
            sd[p + '.conv_module.norm.num_batches_tracked'] = np.asarray(1000, np.int64)
        b = 1.0 / math.sqrt(d)
        sd[p + '.conv_module.pointwise_conv2.weight'] = _uniform(
            seed, p + '.pw2.w', (d, d, 1), b)
        sd[p + '.conv_module.pointwise_conv2.bias'] = _uniform(
            seed, p + '.pw2.b', (d, ), b)
        for n in ('norm_ff', 'norm_mha', 'norm_ff_macaron', 'norm_conv',
                  'norm_final'):
            norm(f'{p}.{n}', d)

    def decoder(prefix, nblocks):
        sd[prefix + '.embed.0.weight'] = _normal(seed, prefix + '.embed',
                                                 (V, d), 1.0)
        sd[prefix + '.embed.1.pe'] = positional_table(d).numpy()
        norm(prefix + '.after_norm', d)
        lin(prefix + '.output_layer', V, d)
        for j in range(nblocks):
            p = f'{prefix}.decoders.{j}'
            attn(p + '.self_attn', d)
            attn(p + '.src_attn', d)
            lin(p + '.feed_forward.w_1', dc['linear_units'], d)
            lin(p + '.feed_forward.w_2', d, dc['linear_units'])
            for n in ('norm1', 'norm2', 'norm3'):
                norm(f'{p}.{n}', d)

    if configs.get('decoder', 'bitransformer') == 'bitransformer':
        decoder('decoder.left_decoder', dc['num_blocks'])
        decoder('decoder.right_decoder', dc.get('r_num_blocks', 0))
    else:
        decoder('decoder', dc['num_blocks'])
    lin('ctc.ctc_lo', V, d)
    if sharpen_ctc:
        sd['ctc.ctc_lo.weight'] = sd['ctc.ctc_lo.weight'] * np.float32(12.0)
        sd['ctc.ctc_lo.bias'] = sd['ctc.ctc_lo.bias'] * np.float32(12.0)
        # Non-blank logits are ~N(0, sigma^2) with sigma = 12/sqrt(3); the
        # expected max of V-1 of them follows the Gumbel asymptotics below.
        # Blank gets a constant logit slightly above it (zero weight row), so
        # roughly 2/3..4/5 of the frames come out blank.
        n = max(V - 1, 2)
        a = math.sqrt(2.0 * math.log(n))
        emax = a - (math.log(math.log(n)) + math.log(4 * math.pi)) / (2 * a) \
            + 0.5772 / a
        sd['ctc.ctc_lo.weight'][0, :] = 0.0
        sd['ctc.ctc_lo.bias'][0] = np.float32(12.0 / math.sqrt(3.0) * emax *
                                              1.10)
    # decoders: sharpen the output layers the same way so rescoring scores
    # separate hypotheses by O(1) rather than by rounding noise.
    for k in list(sd.keys()):
        if k.endswith('output_layer.weight') or k.endswith('output_layer.bias'):
            sd[k] = sd[k] * np.float32(4.0)
    return OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v)))
                       for k, v in sd.items())


def whisper_positional_table(d_model: int, max_len: int = 1500) -> torch.Tensor:
    """The `pe` buffer of WhisperPositionalEncoding
    (wenet/models/transformer/embedding.py:154-163)."""
    inc = np.log(10000) / (d_model // 2 - 1)
    inv = torch.exp(-inc * torch.arange(d_model // 2))
    st = torch.arange(max_len)[:, np.newaxis] * inv[np.newaxis, :]
    return torch.cat([torch.sin(st), torch.cos(st)], dim=1).unsqueeze(0)


def _make_transformer_state_dict(configs, seed, sharpen_ctc):
    """TransformerEncoder (Whisper recipe) + CTC head; names as the reference's
    state_dict (encoder.embed.conv.{0,2}, encoder.encoders.i.{self_attn,
    feed_forward,norm1,norm2}, encoder.after_norm, ctc.ctc_lo)."""
    ec = configs['encoder_conf']
    d, ffn = ec['output_size'], ec['linear_units']
    idim, V = configs['input_dim'], configs['output_dim']
    sd: Dict[str, np.ndarray] = OrderedDict()

    def lin(name, out_f, in_f, bias=True):
        b = 1.0 / math.sqrt(in_f)
        sd[name + '.weight'] = _uniform(seed, name + '.weight', (out_f, in_f), b)
        if bias:
            sd[name + '.bias'] = _uniform(seed, name + '.bias', (out_f, ), b)

    def norm(name, n):
        sd[name + '.weight'] = _normal(seed, name + '.weight', (n, ), 0.1, 1.0)
        sd[name + '.bias'] = _normal(seed, name + '.bias', (n, ), 0.1)

    b = 1.0 / math.sqrt(3 * idim)
    sd['encoder.embed.conv.0.weight'] = _uniform(seed, 'tconv0.w', (d, idim, 3), b)
    sd['encoder.embed.conv.0.bias'] = _uniform(seed, 'tconv0.b', (d, ), b)
    b = 1.0 / math.sqrt(3 * d)
    sd['encoder.embed.conv.2.weight'] = _uniform(seed, 'tconv2.w', (d, d, 3), b)
    sd['encoder.embed.conv.2.bias'] = _uniform(seed, 'tconv2.b', (d, ), b)
    sd['encoder.embed.pos_enc.pe'] = whisper_positional_table(d).numpy().astype(
        np.float32)
    norm('encoder.after_norm', d)
    for i in range(ec['num_blocks']):
        p = f'encoder.encoders.{i}'
        lin(p + '.self_attn.linear_q', d, d)
        lin(p + '.self_attn.linear_k', d, d, bias=bool(ec.get('key_bias', True)))
        lin(p + '.self_attn.linear_v', d, d)
        lin(p + '.self_attn.linear_out', d, d)
        lin(p + '.feed_forward.w_1', ffn, d)
        lin(p + '.feed_forward.w_2', d, ffn)
        norm(p + '.norm1', d)
        norm(p + '.norm2', d)
    lin('ctc.ctc_lo', V, d)
    if sharpen_ctc:
        sd['ctc.ctc_lo.weight'] = sd['ctc.ctc_lo.weight'] * np.float32(12.0)
        sd['ctc.ctc_lo.bias'] = sd['ctc.ctc_lo.bias'] * np.float32(12.0)
        n = max(V - 1, 2)
        a = math.sqrt(2.0 * math.log(n))
        emax = a - (math.log(math.log(n)) + math.log(4 * math.pi)) / (2 * a) \
            + 0.5772 / a
        sd['ctc.ctc_lo.weight'][0, :] = 0.0
        sd['ctc.ctc_lo.bias'][0] = np.float32(12.0 / math.sqrt(3.0) * emax * 1.10)
    return OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v)))
                       for k, v in sd.items())


def make_features(batch: int, num_frames, seed: int = 1234,
                  feat_dim: int = 80) -> Tuple[torch.Tensor, torch.Tensor]:
    """Padded (B, Tmax, F) fbank-like features + int32 lengths, sorted by
    length descending like `padding` (wenet/dataset/processor.py:539).
    `num_frames` is an int (fixed length) or a (lo, hi) range."""
    rng = np.random.Generator(np.random.PCG64([seed, 7]))
    if isinstance(num_frames, (tuple, list)):
        lens = rng.integers(num_frames[0], num_frames[1] + 1, size=batch)
    else:
        lens = np.full((batch, ), int(num_frames))
    lens = np.sort(lens)[::-1].copy()
    tmax = int(lens.max())
    feats = np.zeros((batch, tmax, feat_dim), dtype=np.float32)
    for b in range(batch):
        # log-mel-like: smooth in time, mean ~11, std ~3
        x = rng.standard_normal((int(lens[b]), feat_dim)).astype(np.float32)
        x[1:] = 0.6 * x[:-1] + 0.8 * x[1:]
        feats[b, :lens[b]] = 11.0 + 3.0 * x
    return torch.from_numpy(feats), torch.from_numpy(lens.astype(np.int32))


def make_audio(num_samples: int, seed: int = 1234,
               sample_rate: int = 16000) -> np.ndarray:
    """16 kHz mono fp32 in [-1, 1]: three chirps + coloured noise, peak 0.3
    (SURVEY.md section 8d synthetic-input recipe)."""
    rng = np.random.Generator(np.random.PCG64([seed, 11]))
    t = np.arange(num_samples, dtype=np.float64) / sample_rate
    x = np.zeros(num_samples, dtype=np.float64)
    for _ in range(3):
        f0, f1 = rng.uniform(100, 3000, size=2)
        dur = max(t[-1], 1e-3)
        phase = 2 * np.pi * (f0 * t + 0.5 * (f1 - f0) * t * t / dur)
        x += rng.uniform(0.3, 1.0) * np.sin(phase + rng.uniform(0, 2 * np.pi))
    noise = rng.standard_normal(num_samples)
    noise = np.convolve(noise, np.ones(8) / 8.0, mode='same')
    x += 0.5 * noise
    x *= 0.3 / max(np.abs(x).max(), 1e-9)
    return x.astype(np.float32)


def write_model_dir(path: str, name: str, seed: int = 0) -> str:
    """Write a directory `wenet.load_model` / `wenet_amd.load_model` accepts
    (wenet/cli/model.py:71-92): train.yaml, final.pt, units.txt, global_cmvn."""
    import json
    import yaml
    os.makedirs(path, exist_ok=True)
    configs = make_configs(name)
    sd = make_state_dict(configs, seed)
    # JSON CMVN stats that reproduce the synthetic mean / istd through
    # wenet/utils/cmvn.py:21-43
    mean = sd['encoder.global_cmvn.mean'].double().numpy()
    istd = sd['encoder.global_cmvn.istd'].double().numpy()
    n = 1000
    var = 1.0 / (istd * istd)
    with open(os.path.join(path, 'global_cmvn'), 'w') as f:
        json.dump({'mean_stat': (mean * n).tolist(),
                   'var_stat': ((var + mean * mean) * n).tolist(),
                   'frame_num': n}, f)
    V = configs['output_dim']
    with open(os.path.join(path, 'units.txt'), 'w', encoding='utf8') as f:
        f.write('<blank> 0\n<unk> 1\n<sos/eos> 2\n')
        for i in range(3, V):
            f.write(f'u{i} {i}\n')
    configs['cmvn_conf']['cmvn_file'] = os.path.join(path, 'global_cmvn')
    configs['tokenizer_conf']['symbol_table_path'] = os.path.join(
        path, 'units.txt')
    with open(os.path.join(path, 'train.yaml'), 'w') as f:
        yaml.safe_dump(configs, f)
    torch.save(sd, os.path.join(path, 'final.pt'))
    return path


def peaky_logprobs(batch: int, frames, vocab: int, peak: float, seed: int):
    """Seeded (B, T, V) float32 log-softmax rows with two boosted tokens per frame
    and frequent blanks (token 0), plus int32 lengths in `frames`: inputs for the
    search-only fixtures (prefixes re-merge, biasing phrases match often)."""
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(frames[0], frames[1] + 1, (batch, ), generator=g)
    lens[0] = frames[1]
    T = int(lens.max())
    x = torch.randn(batch, T, vocab, generator=g)
    hot = torch.randint(0, vocab, (batch, T, 2), generator=g)
    x.scatter_add_(2, hot, torch.full(hot.shape, float(peak)))
    x[..., 0] += 1.0
    return torch.log_softmax(x, dim=-1), lens.to(torch.int32)


# ---------------------------------------------------------------------------
# BASELINE.json `configs` as bench.py workloads.  bench.py, the `-m gpu` parity
# tests at these sizes and oracle/gen_golden_bench.py (which runs the REAL
# reference on them) all take the batch from here, so the committed goldens
# (tests/golden/bench_*.npz) are the reference's answer to exactly the batch the
# benchmark times.
BENCH_BEAM = 10
BENCH_FRAMES = (800, 1200)   # 8..12 s of 10 ms frames, mean ~10 s
BENCH_WORKLOADS = {
    'config2': dict(config='aishell_u2pp', batch=32, method='ctc_prefix_beam_search',
                    kw={}, text='BASELINE.json configs[1]: AIShell u2++ conformer '
                    '12L/4head/256d fbank80, batch 32 x ~10 s per GPU (8-12 s '
                    'ragged), ctc_prefix_beam_search beam 10'),
    'config3': dict(config='librispeech_bidecoder_large', batch=64,
                    method='attention_rescoring',
                    kw=dict(ctc_weight=0.5, reverse_weight=0.3),
                    text='BASELINE.json configs[2]: LibriSpeech conformer '
                    'bidecoder-large 12L/8head/512d fbank80, batch 64 x ~10 s per GPU, '
                    'attention_rescoring beam 10, ctc_weight 0.5, reverse_weight 0.3'),
    'config4': dict(config='wenetspeech_u2pp', batch=32,
                    method='ctc_prefix_beam_search',
                    kw=dict(decoding_chunk_size=16, num_decoding_left_chunks=-1),
                    text='BASELINE.json configs[3]: WenetSpeech u2++ conformer '
                    '12L/8head/512d, decoding_chunk_size 16 (chunk-mask streaming), '
                    'batch 32 x ~10 s per GPU, ctc_prefix_beam_search beam 10'),
    'config5': dict(config='whisper_largev3', batch=16, method='ctc_greedy_search',
                    kw={}, frames=(3000, 3000), feat_dim=128,
                    text='BASELINE.json configs[4]: Whisper-large-v3 encoder '
                    '32L/20head/1280d, 128 mel bins, 30 s windows, batch 16 per GPU, '
                    '+ a 307-way CTC head and greedy search'),
}


def make_bench_group(workload: str, group: int = 0):
    """Utterance group `group` of a bench workload: `batch` utterances, seed
    1234 + group.  Group 0 is the whole batch of a 1-GPU run."""
    wl = BENCH_WORKLOADS[workload]
    return make_features(wl['batch'], wl.get('frames', BENCH_FRAMES),
                         seed=1234 + group, feat_dim=wl.get('feat_dim', 80))


def make_bench_batch(workload: str, world: int = 1):
    """Global batch of `bench.py --workload W --gpus world`: groups 0..world-1
    back to back (global index = group * batch + index in group), zero-padded to
    the longest utterance.  Weak scaling: `batch` utterances per GPU."""
    groups = [make_bench_group(workload, g) for g in range(world)]
    if world == 1:
        return groups[0]
    tmax = max(int(f.size(1)) for f, _ in groups)
    feats = torch.zeros((sum(f.size(0) for f, _ in groups), tmax, groups[0][0].size(2)),
                        dtype=torch.float32)
    r = 0
    for f, _ in groups:
        feats[r:r + f.size(0), :f.size(1)] = f
        r += f.size(0)
    return feats, torch.cat([l for _, l in groups])
