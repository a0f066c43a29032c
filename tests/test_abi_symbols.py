"""CPU-only: both shared libraries load without a GPU and export every entry point
their headers (include/*.h) declare; the ctypes signature tables name exactly those."""
import ctypes
import os
import re

from jlm_amd import _lib, lattice

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    with open(os.path.join(REPO, "include", header)) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(jlm_[a-z0-9_]+)\s*\(", text)))


def test_hip_library_exports_header():
    names = _declared("jlm_hip.h")
    assert len(names) >= 12
    lib = ctypes.CDLL(_lib.LIB_PATH)             # loads on a GPU-less box: no HIP call at load time
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.EXPORTS) == names
    assert lib.jlm_abi_version() == 11


def test_host_library_exports_header():
    names = _declared("jlm_host.h")
    lib = lattice.host_lib()
    assert lib is not None, "libjlm_host.so not built (python __graft_entry__.py)"
    for n in names:
        assert hasattr(lib, n), n
