"""The PyTorch-ROCm custom-op boundary: ``torch.ops.jlm.*`` / ``torch.classes.jlm.*`` (csrc/jlm_torch_ops.cpp).

Everything ``jlm_amd`` runs on the device goes through here: the ops take ``torch.Tensor`` arguments (torch owns the
buffers; no raw pointer leaves Python), launch on the current torch HIP stream and raise ``RuntimeError`` with the HIP error
string when a launcher fails.  Underneath sits the C ABI of ``libjlm_hip.so`` (include/jlm_hip.h), which
``jlm_amd/_lib.py`` still binds through ctypes for the kernel unit tests and the developer tools.

There is no CPU fallback: :func:`backend` raises when the extension is not built.  (The CPU-only test-suite installs a
numpy double with the same methods -- tests/fake_hip.py -- through :func:`set_backend`; the product never does.)
"""
import os

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
OPS_PATH = os.path.join(_HERE, "_torch_ops.so")

_backend = None


class HipOps:
    """torch.ops.jlm, loaded once.  Attribute access forwards to the op namespace (``ops.decode_frames(...)``);
    ``Model`` / ``Plan`` construct the custom classes."""

    def __init__(self):
        if not os.path.exists(OPS_PATH) or not os.path.exists(_lib.LIB_PATH):
            raise _lib.JlmHipError(
                "the HIP extension is not built (%s, %s).  Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'`; there is no CPU fallback." % (_lib.LIB_PATH, OPS_PATH))
        import torch
        torch.ops.load_library(OPS_PATH)
        self._ops = torch.ops.jlm
        self.Model = torch.classes.jlm.Model
        self.Plan = torch.classes.jlm.Plan
        if int(self._ops.abi_version()) != 11:
            raise _lib.JlmHipError("libjlm_hip.so ABI version mismatch")

    def __getattr__(self, name):
        return getattr(self._ops, name)


def backend():
    global _backend
    if _backend is None:
        _backend = HipOps()
    return _backend


def set_backend(b):
    """Test hook (tests/fake_hip.py): route the package to another implementation of this interface."""
    global _backend
    _backend = b
