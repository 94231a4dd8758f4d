"""
GPU (-m gpu): exhaustive sweeps of the device math that feeds GAME STATE against the host libm (glibc, what the compiled
reference links), through the self-test hooks of libenv.so (include/procgen_amd.h).

* bigfish's fish radius (reference src/games/bigfish.cpp:84) is a function of one rand01() draw, and rand01() =
  float(u32 / 2^32) takes 83 886 081 distinct values: every one of them is checked.
* sin / cos (bullet and thrust directions in bossfight, caveflyer, ninja, starpilot) are called on FLOAT angles: every
  float with |x| < 1024 is checked.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from helpers import HIP_LIB

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("libm") / "liblibm_sweep.so")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-fPIC", "-shared", os.path.join(HERE, "tools", "libm_sweep.c"), "-lm", "-o", out])
    L = C.CDLL(out)
    L.ref_bigfish_radius.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
    L.count_sincos_mismatches.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_long, C.c_void_p]
    L.collect_sincos_mismatches.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_long, C.c_void_p, C.c_long, C.c_void_p]
    L.ref_sincos_scaled.argtypes = [C.c_void_p, C.c_long, C.c_double, C.c_void_p, C.c_void_p]
    L.count_puff_mismatches.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_long, C.c_float, C.c_float, C.c_int, C.c_void_p]
    return L


@pytest.fixture(scope="module")
def dev():
    L = C.CDLL(HIP_LIB)
    L.procgen_amd_selftest_bigfish_radius.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.procgen_amd_selftest_sincos.argtypes = [C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    L.procgen_amd_selftest_sincos_scaled.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
    return L


def rand01_domain_chunks(chunk=1 << 23):
    """Every value RandGen::rand01 can return (reference src/randgen.cpp:19-23: float(double(u32) / 2^32))."""
    # below 2^-8 a float is finer than 2^-32: only the multiples u * 2^-32, u < 2^24, occur (and are exact)
    for u0 in range(0, 1 << 24, chunk):
        yield (np.arange(u0, min(u0 + chunk, 1 << 24), dtype=np.float64) / 4294967296.0).astype(np.float32)
    # from 2^-8 up every float is a rounding of some u / 2^32, 1.0 included
    lo, hi = int(np.float32(2.0 ** -8).view(np.uint32)), int(np.float32(1.0).view(np.uint32))
    for b0 in range(lo, hi + 1, chunk):
        yield np.arange(b0, min(b0 + chunk, hi + 1), dtype=np.uint32).view(np.float32)


def test_bigfish_radius_equals_host_libm_for_every_rand01_value(dev, host):
    total = bad = 0
    worst = None
    for r in rand01_domain_chunks():
        r = np.ascontiguousarray(r)
        got = np.empty_like(r)
        ref = np.empty_like(r)
        dev.procgen_amd_selftest_bigfish_radius(r.ctypes.data, got.ctypes.data, len(r))
        host.ref_bigfish_radius(r.ctypes.data, ref.ctypes.data, len(r))
        diff = got.view(np.uint32) != ref.view(np.uint32)
        total += len(r)
        bad += int(diff.sum())
        if diff.any() and worst is None:
            i = int(np.argmax(diff))
            worst = (float(r[i]), float(got[i]), float(ref[i]))
    assert total == (1 << 24) + (1 << 26) + 1
    assert bad == 0, f"{bad} of {total} rand01 values give another fish radius than the host libm, e.g. r01, device, host = {worst}"


def test_sin_cos_of_every_float_angle_against_host_libm(dev, host):
    """All floats 0 <= x < 1024 (and a sample of negative ones: both implementations are odd / even by construction).
    As doubles the restated fdlibm algorithm and glibc may differ in the last bit; what the games use is narrowed."""
    chunk = 1 << 24
    hi = int(np.float32(1024.0).view(np.uint32))
    s = np.empty(chunk, np.float64)
    c = np.empty(chunk, np.float64)
    counts = (C.c_long * 4)()
    total = 0
    ranges = [(b0, min(chunk, hi - b0)) for b0 in range(0, hi, chunk)]
    ranges += [(0x80000000 + b0, chunk) for b0 in range(0x3C000000, 0x41000000, 5 * chunk)]  # negative sample, |x| in [2^-7, 8)
    for b0, n in ranges:
        dev.procgen_amd_selftest_sincos(b0, n, s.ctypes.data, c.ctypes.data)
        host.count_sincos_mismatches(s.ctypes.data, c.ctypes.data, b0, n, counts)
        total += n
    ds, dc, fs, fc = list(counts)
    print(f"\\nsin/cos sweep over {total} float angles: double-level mismatches sin {ds} cos {dc}; after narrowing to float sin {fs} cos {fc}")
    assert fs == 0 and fc == 0, f"narrowed to float: {fs} sin and {fc} cos values differ from the host libm ({total} angles)"
    assert ds + dc < total * 2e-2  # last-bit differences in double are expected to be rare


# float(trig(theta) * speed) with the speeds the games use: ninja 1; bossfight .5 (reflected bullets), .5 / .75 (boss bullets,
# easy / hard); starpilot hp_vs * V_SCALE in float arithmetic (game_starpilot.h hp_vs, V_SCALE = 2 / 5, HP_SLOW_V = .5)
SPEEDS = sorted({float(np.float32(v) * np.float32(np.float32(2.0) / np.float32(5.0))) for v in (1.5, 2.0, 1.25, 0.75, 1.0, 0.5)} | {1.0, 0.5, 0.75})


def test_scaled_sin_cos_at_every_angle_where_the_doubles_differ(dev, host):
    """Where device and host sin / cos agree as doubles, every later product agrees.  The float angles (0 <= x < 1024)
    where they differ in the last bit are collected, and the call sites' expression float(trig * speed) is evaluated at
    every one of them for every speed constant of the games: the velocities written into game state are identical."""
    chunk = 1 << 24
    hi = int(np.float32(1024.0).view(np.uint32))
    s = np.empty(chunk, np.float64)
    c = np.empty(chunk, np.float64)
    cap = 16 << 20
    cand = np.zeros(cap, np.uint32)
    count = C.c_long(0)
    for b0 in range(0, hi, chunk):
        n = min(chunk, hi - b0)
        dev.procgen_amd_selftest_sincos(b0, n, s.ctypes.data, c.ctypes.data)
        host.collect_sincos_mismatches(s.ctypes.data, c.ctypes.data, b0, n, cand.ctypes.data, cap, C.byref(count))
    k = count.value
    assert 0 < k <= cap
    cand = np.ascontiguousarray(cand[:k])
    cand = np.concatenate([cand, cand | np.uint32(0x80000000)])  # and their negatives
    gs, gc, rs, rc = (np.empty(len(cand), np.float32) for _ in range(4))
    for speed in SPEEDS:
        dev.procgen_amd_selftest_sincos_scaled(cand.ctypes.data, len(cand), speed, gs.ctypes.data, gc.ctypes.data)
        host.ref_sincos_scaled(cand.ctypes.data, len(cand), speed, rs.ctypes.data, rc.ctypes.data)
        bad = int((gs.view(np.uint32) != rs.view(np.uint32)).sum() + (gc.view(np.uint32) != rc.view(np.uint32)).sum())
        assert bad == 0, f"speed {speed}: {bad} of {2 * len(cand)} velocity components differ from the host libm"
    print(f"\\n{k} float angles with a last-bit difference in double; float(trig * speed) identical for all of them at speeds {SPEEDS}")


def test_caveflyer_exhaust_puff_position_at_every_angle_where_the_doubles_differ(dev, host):
    """The one trig call site whose result reaches serialized state through a position-dependent operand (reference
    src/games/caveflyer.cpp:275; game_caveflyer.h set_action_xy): float(x - r * trig(theta)), r = the agent's radius (0.4f in every mode,
    rx = ry), x = its position.  At every float angle 0 <= |theta| < 1024 where the device's sin or cos differs from the host libm's in the
    last bit of the double, the expression is evaluated on the device's doubles and on the host's for 2048 positions across the largest
    world (60 cells, memory mode) and the 65 floats around the wall contact x = r (tests/tools/libm_sweep.c count_puff_mismatches).
    NOT closed to zero: a last-bit difference of the double survives the subtraction and the narrowing when x - r trig lands within
    ~1e-17 of a float rounding boundary -- measured in round 4: 1 of 1.86e10 evaluated (angle, position) pairs, i.e. with 0.58 % of the
    angles affected about 3e-13 per puff (a cosmetic entity that lives four steps).  Only glibc's own sin / cos (IBM accurate math
    tables, not on this machine in source form) would close it; the test pins the rate."""
    chunk = 1 << 24
    hi = int(np.float32(1024.0).view(np.uint32))
    s = np.empty(chunk, np.float64)
    c = np.empty(chunk, np.float64)
    counts = (C.c_long * 4)()
    for b0 in range(0, hi, chunk):
        n = min(chunk, hi - b0)
        dev.procgen_amd_selftest_sincos(b0, n, s.ctypes.data, c.ctypes.data)
        host.count_puff_mismatches(s.ctypes.data, c.ctypes.data, b0, n, 0.4, 60.0, 2048, counts)
    angles, products, evals, bad = list(counts)
    print(f"\\ncaveflyer puff: {angles} angles with a last-bit difference, {products} differing products r * trig, {evals} positions evaluated, {bad} differing")
    assert angles > 0 and evals > 1e10 and bad <= 8, f"{bad} of {evals} puff positions differ: the rate measured in round 4 was 1 in 1.86e10"
