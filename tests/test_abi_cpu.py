"""CPU-side checks of the C-ABI boundary: the shared library loads without a GPU and exports every symbol that
include/mrblip_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mr-blip_amd", "csrc", "libmrblip_hip.so")
HDR = os.path.join(ROOT, "include", "mrblip_hip.h")


def _declared():
    txt = open(HDR).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mrblip_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(LIB):
        import sys
        sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd", "csrc"))
        import build
        build.build()
    lib = ctypes.CDLL(LIB)
    names = _declared()
    assert len(names) >= 24
    for n in names:
        assert hasattr(lib, n), n
    lib.mrblip_abi_version.restype = ctypes.c_int
    assert lib.mrblip_abi_version() == 1


def test_python_bindings_cover_the_header():
    from mrblip import ops
    assert sorted(ops.EXPORTS) == _declared()


def test_argument_validation_needs_no_gpu():
    from mrblip import ops
    lib = ops._lib
    # K % 64 != 0 must be rejected before any launch
    rc = ops._gemm(None, 8, None, 8, None, 0, None, 0, 4, 8, 10, None, 8, 0, None, 0, None, None, 0, 0, 0, None, 0, 0.0, 0, None)
    assert rc == -1 and b"K%64" in lib.mrblip_last_error()
