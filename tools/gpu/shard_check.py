"""The body of tests/test_gpu_parity_at_scale.py::test_one_handle_sharded_over_devices_equals_the_single_device_handle, section by
section with prints and without torch / pytest (a quick look on a GPU box: python tools/gpu/shard_check.py)."""
import os, sys, ctypes as C, traceback
os.environ["PROCGEN_AMD_FAKE_DEVICES"] = "1"
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (REPO, os.path.join(REPO, "tests"), os.path.join(REPO, "tests", "tools"), os.path.join(REPO, "oracle")):
    sys.path.insert(0, p)
from helpers import action_stream, rollout
from procgen_amd import ProcgenGym3Env


def make_env(n, game, **kw):
    kw.setdefault("rand_seed", 23)
    return ProcgenGym3Env(n, game, **kw)


def section(name, f):
    try:
        f()
        print("ok  :", name, flush=True)
    except Exception:
        print("FAIL:", name, flush=True)
        traceback.print_exc()


n, steps = 96, 90
acts = action_stream(n, steps, seed=12)


def s1():
    one = rollout(make_env(n, "starpilot"), acts, keep_frames=True)
    for G in (2, 4):
        got = rollout(make_env(n, "starpilot", extra_options={"num_devices": G}), acts, keep_frames=True)
        for k in one:
            if not np.array_equal(one[k], got[k]):
                bad = np.argwhere(np.asarray(one[k]) != np.asarray(got[k]))
                raise AssertionError(f"G={G} {k}: first mismatch at {bad[0]} of {len(bad)}")


names = ["coinrun", "bigfish", "maze"]


def s2():
    a = rollout(make_env(n, ",".join(names)), acts)
    b = rollout(make_env(n, ",".join(names), extra_options={"num_devices": 2}), acts)
    for k in a:
        if not np.array_equal(a[k], b[k]):
            bad = np.argwhere(np.asarray(a[k]) != np.asarray(b[k]))
            raise AssertionError(f"joint {k}: first mismatch at {bad[0]} of {len(bad)}")


def s3():
    import state_parse
    env = make_env(n, ",".join(names), extra_options={"num_devices": 2})
    sts = env.get_state()
    for e in (0, 47, 48, 95):
        st = state_parse.parse_state(sts[e])
        assert st["game_name"] == names[e % 3] and st["game_n"] == e, (e, st["game_name"], st["game_n"])
    env2 = make_env(n, ",".join(names), rand_seed=5, extra_options={"num_devices": 2})
    env2.set_state(sts)
    got = env2.get_state()
    diff = [e for e in range(n) if got[e] != sts[e]]
    if diff:
        e = diff[0]
        a, b = np.frombuffer(sts[e], np.uint8), np.frombuffer(got[e], np.uint8)
        where = np.nonzero(a[: min(len(a), len(b))] != b[: min(len(a), len(b))])[0]
        raise AssertionError(f"states differ for envs {diff[:8]} ({len(diff)}); env {e}: lengths {len(a)} {len(b)}, first differing byte {where[:6]}")


section("sharded starpilot = single device", s1)
section("joint x shards", s2)
section("get_state / set_state with global indices", s3)
