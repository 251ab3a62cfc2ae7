import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")
    config.addinivalue_line("markers", "slow: long CPU test")


def has_reference():
    return os.path.isdir(os.path.join(
        os.environ.get("WENET_REFERENCE_ROOT", "/root/reference"), "wenet"))


needs_reference = pytest.mark.skipif(
    not has_reference(), reason="reference tree not present (GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a host without a GPU (or without the built library)
    skips the `gpu`-marked tests instead of dying in the HIP runtime."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    lib = os.path.join(ROOT, 'wenet_amd', 'libwenet_amd.so')
    if has_gpu and os.path.exists(lib):
        return
    reason = 'no GPU' if not has_gpu else 'libwenet_amd.so not built'
    skip = pytest.mark.skip(reason=reason)
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session', autouse=True)
def _tune_from_env():
    """WN_TUNE="key=value,key=value": wn_tune_set knobs for a whole test session (A/B of a
    kernel variant under the parity tests); unset = the defaults."""
    spec = os.environ.get('WN_TUNE', '')
    if spec:
        from wenet_amd import _lib
        for kv in filter(None, spec.split(',')):
            k, v = kv.split('=')
            _lib.check(_lib.lib().wn_tune_set(k.encode(), int(v)), 'tune')
    yield
