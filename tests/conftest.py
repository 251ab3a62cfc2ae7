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
