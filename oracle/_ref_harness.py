"""Import shim for the *reference* WeNet tree (test infrastructure only).

Lets `import wenet` succeed from /root/reference in a container that lacks
torchaudio / librosa / langid / openai-whisper.  It is
used ONLY by oracle/gen_golden.py and by the CPU tests that pin the oracle
against the real reference; nothing in the product path (wenet_amd/) imports it,
and it is never used on the GPU box (where /root/reference does not exist).
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types
import typing

REFERENCE_ROOT = os.environ.get("WENET_REFERENCE_ROOT", "/root/reference")
_MISSING = ("torchaudio", "librosa", "langid", "whisper", "jieba",
            "tensorboardX", "deepspeed", "textgrid")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "wenet"))


class _Stub(types.ModuleType):

    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)

        class _Any:

            def __init__(s, *a, **k):
                pass

            def __call__(s, *a, **k):
                return _Any()

            def __getattr__(s, k):
                return _Any()

            @staticmethod
            def from_modelstring(*a, **k):
                return _Any()

        return _Any


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):

    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in _MISSING:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Stub(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, m):
        pass


_installed = False


def install():
    """Make `import wenet` (the reference) work.  Idempotent."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.path.insert(0, REFERENCE_ROOT)
    sys.meta_path.insert(0, _Finder())
    import torchaudio.compliance.kaldi as _tk
    _tk.Tuple = typing.Tuple
    import whisper.tokenizer as _wt
    _wt.LANGUAGES = {"en": "english", "zh": "chinese"}
    import torch  # noqa: F401
    import torch.nn.modules.conv as _c
    for n in ("Union", "Optional"):
        if not hasattr(_c, n):
            setattr(_c, n, getattr(typing, n))
    _installed = True
