"""CPU: the C-ABI library loads and exports every entry point include/*.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

from helpers import HIP_LIB, REPO


def declared_symbols():
    syms = []
    for hdr in ("libenv.h", "procgen_amd.h"):
        src = open(os.path.join(REPO, "include", hdr)).read()
        syms += re.findall(r"LIBENV_API\s+[\w\s\*]+?\b(\w+)\s*\(", src)
    return sorted(set(syms))


def test_headers_declare_the_reference_entry_points():
    syms = declared_symbols()
    for s in ("libenv_version", "libenv_make", "libenv_get_tensortypes", "libenv_set_buffers", "libenv_observe", "libenv_act",
              "libenv_close", "get_state", "set_state"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    if not os.path.exists(HIP_LIB):
        pytest.skip("libenv.so not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(HIP_LIB)
    for s in declared_symbols():
        assert hasattr(lib, s), f"libenv.so does not export {s}"
    lib.libenv_version.restype = ctypes.c_int
    assert lib.libenv_version() == 1


def test_struct_layout_of_the_ctypes_mirror():
    from procgen_amd import libenv

    assert ctypes.sizeof(libenv.TensorType) == 128 + 4 + 4 + 64 + 4 + 4 + 4
    assert ctypes.sizeof(libenv.Option) == 128 + 4 + 4 + 8
    assert ctypes.sizeof(libenv.Buffers) == 40
