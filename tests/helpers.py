"""Shared helpers of the parity tests."""
import os
import zlib

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_LIB = os.path.join(REPO, "procgen_amd", "csrc", "build", "libenv.so")


def action_stream(n, steps, seed=0):
    """The reference's determinism protocol (reference procgen/env_test.py:36-44)."""
    rng = np.random.RandomState(seed)
    return [rng.randint(0, 15, size=(n,), dtype=np.int32) for _ in range(steps)]


def rollout(env, actions, keep_frames=False):
    """Returns dict of per-step arrays: rew, first, info x3, crc (and frames)."""
    out = {k: [] for k in ("rew", "first", "prev_level_seed", "prev_level_complete", "level_seed", "crc")}
    frames = []
    for t in range(len(actions) + 1):
        rew, ob, first = env.observe()
        info = env.info_arrays()
        out["rew"].append(np.array(rew, copy=True))
        out["first"].append(np.asarray(first).astype(np.uint8))
        for k in ("prev_level_seed", "prev_level_complete", "level_seed"):
            out[k].append(np.array(info[k], copy=True))
        rgb = ob["rgb"]
        out["crc"].append(np.array([zlib.crc32(rgb[e].tobytes()) for e in range(rgb.shape[0])], dtype=np.uint32))
        if keep_frames:
            frames.append(rgb.copy())
        if t < len(actions):
            env.act(actions[t])
    res = {k: np.array(v) for k, v in out.items()}
    if keep_frames:
        res["frames"] = np.array(frames)
    return res


def assert_rollouts_equal(a, b, what=""):
    for k in ("rew", "first", "prev_level_seed", "prev_level_complete", "level_seed", "crc"):
        if not np.array_equal(a[k], b[k]):
            bad = np.argwhere(a[k] != b[k])
            raise AssertionError(f"{what}: {k} differs first at (step, env) = {tuple(bad[0])}; {len(bad)} mismatches")


def hip_memcpy_dtoh(dev_ptr, nbytes):
    """Reads device memory through the HIP runtime (ctypes), for the zero-copy extension hook tests."""
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")
    buf = np.empty(nbytes, dtype=np.uint8)
    rc = hip.hipMemcpy(C.c_void_p(buf.ctypes.data), C.c_void_p(dev_ptr), C.c_size_t(nbytes), C.c_int(2))
    assert rc == 0, f"hipMemcpy failed: {rc}"
    return buf


# tests/golden/option_matrix.npz (make_golden.py options): option sets by name
OPTION_SETS = {
    "no_backgrounds": dict(use_backgrounds=False),
    "no_center_agent": dict(center_agent=False),
    "restrict_themes": dict(restrict_themes=True),
    "two_levels": dict(num_levels=2, start_level=5),
    "sequential_levels": dict(use_sequential_levels=True, num_levels=3),
    "monochrome": dict(use_monochrome_assets=True),
    "vel_info": dict(paint_vel_info=True),
}


def check_against_option_matrix(g, make, pairs):
    """make(game, n, **options) -> env; compares rew / first / level_seed / frame CRCs with the fixture for the given (game, set) pairs."""
    for game, name in pairs:
        n = g[f"{game}/{name}/rew"].shape[1]
        steps = g[f"{game}/{name}/rew"].shape[0] - 1
        got = rollout(make(game, n, **OPTION_SETS[name]), action_stream(n, steps, seed=1))
        for k in ("rew", "first", "level_seed", "crc"):
            assert np.array_equal(got[k], g[f"{game}/{name}/{k}"]), (game, name, k)

def check_against_generated_assets_fixture(g, make, games):
    """make(game, n, use_generated_assets=True) -> env; rew / first / level_seed / frame CRCs of tests/golden/generated_assets.npz
    (compiled reference, make_golden.py generated)."""
    for game in games:
        n = g[f"{game}/rew"].shape[1]
        steps = g[f"{game}/rew"].shape[0] - 1
        got = rollout(make(game, n, use_generated_assets=True), action_stream(n, steps, seed=2), keep_frames=f"{game}/frames0" in g.files)
        for k in ("rew", "first", "level_seed"):
            assert np.array_equal(got[k], g[f"{game}/{k}"]), (game, k)
        bad = np.argwhere(got["crc"] != g[f"{game}/crc"])
        if len(bad) and f"{game}/frames0" in g.files:
            t = int(bad[0][0])
            diff = (np.asarray(got["frames"][t][0]) != g[f"{game}/frames0"][t]).any(-1).sum()
            assert False, (game, "frame CRC", len(bad), "first at step", t, "env-0 pixels differing there:", int(diff))
        assert len(bad) == 0, (game, "frame CRC", len(bad), bad[:4])
