"""
pg_math.h (the libm functions of the reference's STATE path, restated for the device) against the host libm --
the library the compiled reference links.  Runs the host build of the same header (tests/emu).
"""
import ctypes as C
import ctypes.util

import numpy as np

from emu import emu_harness


def _libm_atan2f(y, x):
    libm = C.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    libm.atan2f.restype = C.c_float
    libm.atan2f.argtypes = [C.c_float, C.c_float]
    return np.array([libm.atan2f(float(a), float(b)) for a, b in zip(y, x)], dtype=np.float32)


def test_atan2f_matches_host_libm_bit_for_bit():
    L = emu_harness.lib()
    L.emu_atan2f_array.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    rng = np.random.RandomState(7)
    parts = [
        (rng.uniform(-20, 20, 60000), rng.uniform(-20, 20, 60000)),            # entity-to-agent offsets in world units
        (rng.uniform(-1, 1, 20000) * 1e-3, rng.uniform(-1, 1, 20000)),          # nearly horizontal
        (rng.uniform(-1, 1, 20000), rng.uniform(-1, 1, 20000) * 1e-3),          # nearly vertical
    ]
    bits = rng.randint(0, 2 ** 32, size=(2, 40000), dtype=np.uint64).astype(np.uint32)  # arbitrary bit patterns (inf, nan, denormals)
    parts.append((bits[0].view(np.float32), bits[1].view(np.float32)))
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, 0.8, -0.8, -6.99382e-8, 1e-30], dtype=np.float32)
    sy, sx = np.meshgrid(special, special)
    parts.append((sy.ravel(), sx.ravel()))
    y = np.concatenate([np.asarray(p[0], dtype=np.float32) for p in parts])
    x = np.concatenate([np.asarray(p[1], dtype=np.float32) for p in parts])
    out = np.zeros_like(y)
    L.emu_atan2f_array(y.ctypes.data, x.ctypes.data, out.ctypes.data, len(y))
    ref = _libm_atan2f(y, x)
    both_nan = np.isnan(out) & np.isnan(ref)
    same = (out.view(np.uint32) == ref.view(np.uint32)) | both_nan
    assert same.all(), f"{(~same).sum()} of {len(y)} differ, first: y={y[~same][0]!r} x={x[~same][0]!r}"


def test_double_atan2_narrowed_to_float_matches_host_libm():
    """pg_atan2_d (get_theta, reference src/basic-abstract-game.cpp:233-238: float offsets promoted to double, result
    narrowed to float): within 1 ulp of glibc's correctly rounded atan2 in double, identical after the narrowing."""
    L = emu_harness.lib()
    L.emu_atan2d_array.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    rng = np.random.RandomState(11)
    y = np.concatenate([rng.uniform(-64, 64, 400000), rng.uniform(-1, 1, 100000) * 1e-4, rng.uniform(-64, 64, 100000), [0.0, -0.0, 3.0, -3.0, 0.0, 1e30]])
    x = np.concatenate([rng.uniform(-64, 64, 400000), rng.uniform(-64, 64, 100000), rng.uniform(-1, 1, 100000) * 1e-4, [1.0, -1.0, 0.0, 0.0, -0.0, -1e30]])
    y = y.astype(np.float32).astype(np.float64)  # the game's inputs are float differences
    x = x.astype(np.float32).astype(np.float64)
    out = np.zeros_like(y)
    L.emu_atan2d_array(y.ctypes.data, x.ctypes.data, out.ctypes.data, len(y))
    ref = np.arctan2(y, x)  # numpy calls the host libm
    assert np.array_equal(out.astype(np.float32).view(np.uint32), ref.astype(np.float32).view(np.uint32))
    ulp = np.abs(out - ref) / np.maximum(np.spacing(np.abs(ref)), 5e-324)
    assert ulp.max() <= 1.0


def test_double_sin_cos_narrowed_to_float_match_host_libm():
    """pg_sin_d / pg_cos_d (rotated sprites, bullet and thrust directions; their callers narrow to float or truncate to
    pixels / 16.16 coefficients): within 1 ulp of the host libm in double, identical after the narrowing."""
    L = emu_harness.lib()
    L.emu_sincos_array.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    rng = np.random.RandomState(13)
    x = np.concatenate([
        rng.uniform(-3.2, 3.2, 300000).astype(np.float32).astype(np.float64),                      # angles from atan2f / rand01 * 2 pi
        rng.uniform(-4000, 4000, 300000).astype(np.float32).astype(np.float64),                    # accumulated rotations
        0.017453292519943295769 * rng.uniform(-2e5, 2e5, 300000).astype(np.float32).astype(np.float64),  # QTransform::rotate: deg2rad * a
        rng.uniform(-1e-3, 1e-3, 50000), np.array([0.0, -0.0, np.pi / 4, -np.pi / 4, np.pi / 2, np.pi, 1.5707963267948966, 8.0e5]),
    ])
    s = np.zeros_like(x)
    c = np.zeros_like(x)
    L.emu_sincos_array(x.ctypes.data, s.ctypes.data, c.ctypes.data, len(x))
    for got, ref in ((s, np.sin(x)), (c, np.cos(x))):
        assert np.array_equal(got.astype(np.float32).view(np.uint32), ref.astype(np.float32).view(np.uint32))
        ulp = np.abs(got - ref) / np.maximum(np.spacing(np.abs(ref)), 5e-324)
        assert ulp.max() <= 1.0
