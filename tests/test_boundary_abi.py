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


@pytest.mark.parametrize("n,chunks,first_pct", [(64, 2, 75), (4095, 2, 75), (4096, 2, 75), (65536, 2, 75), (65536, 2, 50), (65536, 1, 75), (12288, 3, 75), (70000, 2, 75), (5000, 4, 0)])
def test_render_launch_order_is_a_permutation_of_every_launch_chunk(n, chunks, first_pct):
    """PROCGEN_AMD_RENDER_ORDER (host code, no device): a chunk's render kernel is ordered behind that chunk's step kernel only, so the
    workgroup -> env map must keep every launch chunk's env range (the ranges of launch_game) to itself; and the workgroups one XCD
    gets (j mod 8) draw few distinct background images at a time."""
    import numpy as np

    if not os.path.exists(HIP_LIB):
        pytest.skip("libenv.so not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(HIP_LIB)
    lib.procgen_amd_selftest_render_order.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    bg = np.random.RandomState(n).randint(0, 62, size=n).astype(np.int32)
    out = np.full(n, -1, np.int32)
    lib.procgen_amd_selftest_render_order(bg.ctypes.data, n, chunks, first_pct, out.ctypes.data)
    # the launch chunks as kernels.hip cuts them (first_chunk_envs / chunk_envs_for): whole tiles of 64 envs, one chunk below 4096 envs
    if n < 4096 or chunks <= 1:
        bounds = [0, n]
    else:
        nchunk = min(chunks, 8)
        first = (n * first_pct // 100) // 64 * 64 if (nchunk == 2 and 0 < first_pct < 100) else 0
        if 0 < first < n:
            bounds = [0, first, n]
        else:
            per = (((n + nchunk - 1) // nchunk) + 63) // 64 * 64
            bounds = [b for b in range(0, n, per)] + [n]
    for b, e in zip(bounds[:-1], bounds[1:]):
        assert sorted(out[b:e].tolist()) == list(range(b, e)), (b, e)
    b, e = bounds[0], bounds[1]
    if e - b >= 16384:  # locality: 64 consecutive workgroups of one XCD see a handful of images, not the ~40 a random order shows
        xcd0 = bg[out[b:e:8]]
        windows = [len(set(xcd0[k:k + 64].tolist())) for k in range(0, len(xcd0) - 64, 64)]
        assert np.median(windows) <= 3, windows[:20]
