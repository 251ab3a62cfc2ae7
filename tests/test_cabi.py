"""CPU checks of the drop-in boundary: libwenet_amd.so loads without a GPU and
exports every symbol include/wenet_amd.h declares (no compute calls here), the
Python binding lists the same set, and argument validation that needs no device
fails with the documented status / message."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'wenet_amd.h')


def _declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(wn_[a-z0-9_]+)\s*\(', src)))


def test_header_declares_the_documented_entry_points():
    syms = _declared_symbols()
    for s in ('wn_model_create', 'wn_model_destroy', 'wn_fbank', 'wn_encode',
              'wn_ctc_logprobs', 'wn_ctc_greedy_search',
              'wn_ctc_prefix_beam_search', 'wn_attention_rescoring',
              'wn_last_error'):
        assert s in syms


def test_library_exports_every_declared_symbol():
    from wenet_amd import _lib, build
    build.build(force=False, verbose=False)
    assert os.path.exists(_lib.LIB_PATH)
    L = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 18
    for s in declared:
        assert hasattr(L, s), f'{s} declared in include/wenet_amd.h but not exported'
    assert sorted(_lib.EXPORTS) == declared, \
        'wenet_amd/_lib.py EXPORTS and include/wenet_amd.h disagree'
    assert b'gfx950' in L.wn_version()


def test_argument_validation_without_a_device():
    from wenet_amd import _lib
    L = _lib.lib()
    # null arguments are rejected before any HIP call
    assert L.wn_model_create(None, None, 0, 0, None) == -1
    assert b'null' in L.wn_last_error()
    assert L.wn_tune_set(b'no_such_knob', 1) == -1
    assert b'unknown key' in L.wn_last_error()
    assert L.wn_tune_set(b'gemm_x6', 1) == 0
    # variants that leave out parts of a kernel (wrong results by design) exist in WN_ABLATION
    # measurement builds only: the product library refuses their values
    if L.wn_tune_set(b'ffn_x6f_ring', 3) != 0:          # (a product build)
        assert L.wn_tune_set(b'ffn_x6f_var', 8706) == -1
        assert b'WN_ABLATION' in L.wn_last_error()
        assert L.wn_tune_set(b'x6_probe', 2) == -1
        assert L.wn_tune_set(b'ffn_x6f_var', 0) == 0 and L.wn_tune_set(b'x6_probe', 0) == 0


def test_tune_table_defaults_and_process_default_round_trip():
    """wn_tune_set writes the process default, wn_tune_get(NULL, ...) reads it; every key of
    csrc/tune.h's table is reachable and starts at its documented default."""
    import ctypes
    import re
    from wenet_amd import _lib
    L = _lib.lib()
    table = open(os.path.join(ROOT, 'wenet_amd', 'csrc', 'tune.h')).read()
    keys = re.findall(r'^\s*X\((\w+), (-?\d+)\)', table, re.M)
    assert len(keys) >= 25 and ('ffn_x6f', '1') in keys
    v = ctypes.c_int32(0)
    for name, dflt in keys:
        assert L.wn_tune_get(None, name.encode(), ctypes.byref(v)) == 0, name
        assert v.value == int(dflt), (name, v.value, dflt)
    assert L.wn_tune_get(None, b'no_such_knob', ctypes.byref(v)) == -1
    assert L.wn_tune_set(b'attn_bf16_nw', 4) == 0
    assert L.wn_tune_get(None, b'attn_bf16_nw', ctypes.byref(v)) == 0 and v.value == 4
    assert L.wn_tune_set(b'attn_bf16_nw', 0) == 0
    # INT32_MIN is the per-handle "inherit" marker, not a process default
    assert L.wn_tune_set(b'attn_bf16_nw', -2 ** 31) == -1
    assert L.wn_model_tune_set(None, b'attn_bf16_nw', 1) == -1


def test_product_path_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under wenet_amd/ may import,
    call or link it."""
    pkg = os.path.join(ROOT, 'wenet_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(dirpath, f), errors='replace').read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', text, re.M), f
                assert 'wenet_oracle' not in text, f


def test_model_refuses_cpu_device():
    import torch
    from wenet_amd import synthetic as S
    from wenet_amd.model import ASRModel
    configs = S.make_configs('tiny_sym')
    with pytest.raises(RuntimeError):
        ASRModel(configs, {}, device='cpu')
