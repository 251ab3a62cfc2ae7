"""Per-handle tuning knobs (csrc/tune.h): wn_model_tune_set overrides one handle, wn_tune_set the
process default; the effective set is resolved per call and per thread.  No counterpart in the
reference (its kernels are chosen by torch's dispatcher)."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


def _enc(model, feats, lens):
    out, _ = model._forward_encoder(feats.cuda(), lens, -1, -1)
    return out.cpu()


def test_an_override_moves_one_handle_and_only_that_handle():
    from gpu_util import cached_model
    from wenet_amd import _lib, synthetic as S
    L = _lib.lib()
    _, _, model = cached_model('aishell_u2pp', 0)
    other = model.clone()
    feats, lens = S.make_features(32, (800, 1200), seed=83)
    dflt = _enc(model, feats, lens)
    assert torch.equal(dflt, _enc(other, feats, lens))
    try:
        # the v_mfma_f32 kernels as the PROCESS default: the reference for "knob off"
        _lib.check(L.wn_tune_set(b'gemm_x6', 0), 'tune')
        off = _enc(model, feats, lens)
    finally:
        L.wn_tune_set(b'gemm_x6', 1)
    assert not torch.equal(off, dflt)          # (different arithmetic order: the knob is visible)
    try:
        assert other.tune('gemm_x6', 0) == 0
        assert model.tune('gemm_x6') == 1
        assert torch.equal(_enc(other, feats, lens), off)      # the override ...
        assert torch.equal(_enc(model, feats, lens), dflt)     # ... of that handle only
        twin = other.clone()                                    # clones carry the overrides
        assert twin.tune('gemm_x6') == 0
        assert torch.equal(_enc(twin, feats, lens), off)
        # an override outranks the process default in both directions
        _lib.check(L.wn_tune_set(b'gemm_x6', 0), 'tune')
        assert model.tune('gemm_x6', 1) == 1
        assert torch.equal(_enc(model, feats, lens), dflt)
        L.wn_tune_set(b'gemm_x6', 1)
        assert model.tune('gemm_x6', 'inherit') == 1
    finally:
        L.wn_tune_set(b'gemm_x6', 1)
        other.tune('gemm_x6', 'inherit')
        model.tune('gemm_x6', 'inherit')
    assert other.tune('gemm_x6') == 1
    assert torch.equal(_enc(other, feats, lens), dflt)


def test_two_threads_two_handles_two_kernel_forms():
    """Each host thread drives its own handle (the library's threading contract); the one with
    the override keeps its form while the other runs the default beside it."""
    from gpu_util import cached_model
    from wenet_amd import _lib, synthetic as S
    L = _lib.lib()
    _, _, model = cached_model('aishell_u2pp', 0)
    other = model.clone()
    feats, lens = S.make_features(32, (800, 1200), seed=89)   # (the on-chip FFN's batch size)
    dflt = _enc(model, feats, lens)
    try:
        _lib.check(L.wn_tune_set(b'ffn_x6f', 0), 'tune')
        off = _enc(model, feats, lens)
    finally:
        L.wn_tune_set(b'ffn_x6f', 1)
    assert not torch.equal(off, dflt)
    out = {}
    errs = []

    def run(name, m, want):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(6):
                    got = _enc(m, feats, lens)
                    if not torch.equal(got, want):
                        errs.append(name)
            out[name] = True
        except Exception as e:       # noqa: BLE001
            errs.append(f'{name}: {e!r}')

    try:
        other.tune('ffn_x6f', 0)
        th = [threading.Thread(target=run, args=('default', model, dflt)),
              threading.Thread(target=run, args=('override', other, off))]
        for t in th:
            t.start()
        for t in th:
            t.join()
    finally:
        other.tune('ffn_x6f', 'inherit')
    assert not errs, errs
    assert out == {'default': True, 'override': True}
