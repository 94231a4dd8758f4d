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


@pytest.mark.parametrize("count", [1, 7, 8, 17, 64, 100, 700, 4095, 4096, 5000, 49152, 70000])
def test_render_order_slot_is_a_permutation_for_any_count(count):
    """shard_map.h render_order_slot (the device sort's position -> launch slot map): a bijection of [0, count) for every count (round 5's
    form was one only for multiples of 8 and left up to 7 envs of a chunk with stale frames), and XCD x = slot % 8 takes a contiguous
    eighth of the sorted sequence, the shares differing by at most one."""
    import numpy as np

    if not os.path.exists(HIP_LIB):
        pytest.skip("libenv.so not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(HIP_LIB)
    out = np.full(count, -1, dtype=np.int32)
    lib.procgen_amd_selftest_render_order_slots(ctypes.c_int(count), out.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    assert np.array_equal(np.sort(out), np.arange(count)), "not a permutation"
    xcd = out % 8
    shares = np.bincount(xcd, minlength=8)
    assert shares.max() - shares.min() <= 1
    for x in range(8):
        pos = np.nonzero(xcd == x)[0]
        if len(pos):
            assert pos[-1] - pos[0] + 1 == len(pos), "an XCD's share of the sorted sequence is contiguous"
            assert np.array_equal(out[pos] // 8, np.arange(len(pos))), "and is drawn in sorted order"
