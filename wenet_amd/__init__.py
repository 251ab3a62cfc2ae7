"""wenet_amd: MI355X-native Conformer-ASR inference path behind WeNet's API.

Public surface mirrors the reference (wenet/__init__.py:1, wenet/cli/model.py):
``load_model``, plus the search free functions and ``DecodeResult``.
The compute path is hand-written HIP for gfx950 in ``libwenet_amd.so`` (C-ABI in
include/wenet_amd.h); it is loaded lazily and there is NO CPU fallback: using a
model without the library raises.
"""
__all__ = ["load_model", "ASRModel", "DecodeResult"]


def __getattr__(name):
    if name in ("load_model", "ASRModel"):
        from wenet_amd import model as _m
        return getattr(_m, name)
    if name == "DecodeResult":
        from wenet_amd.search import DecodeResult
        return DecodeResult
    raise AttributeError(name)
