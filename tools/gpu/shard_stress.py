"""Round 4: the body of test_one_handle_sharded_over_devices_equals_the_single_device_handle in a loop, with the first
mismatch described (which key, step, env, how many values), so that a rare failure says where it is.
  python tools/gpu/shard_stress.py LOOPS [TAG]      (PROCGEN_AMD_HOST_THREADS from the environment)
Exit code = number of failing sections."""
import os, sys, time, traceback
os.environ["PROCGEN_AMD_FAKE_DEVICES"] = "1"
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (REPO, os.path.join(REPO, "tests"), os.path.join(REPO, "tests", "tools"), os.path.join(REPO, "oracle")):
    sys.path.insert(0, p)
from helpers import action_stream, rollout
from procgen_amd import ProcgenGym3Env

loops = int(sys.argv[1]) if len(sys.argv) > 1 else 10
tag = sys.argv[2] if len(sys.argv) > 2 else "-"
n, steps = 96, 90
acts = action_stream(n, steps, seed=12)
names = ["coinrun", "bigfish", "maze"]


def make_env(n, game, **kw):
    kw.setdefault("rand_seed", 23)
    return ProcgenGym3Env(n, game, **kw)


def compare(a, b, what):
    for k in a:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        if not np.array_equal(x, y):
            bad = np.argwhere(x != y)
            steps_bad = sorted(set(int(i[0]) for i in bad))
            envs_bad = sorted(set(int(i[1]) for i in bad)) if bad.shape[1] > 1 else []
            raise AssertionError(f"{what} {k}: {len(bad)} values differ; steps {steps_bad[:10]} envs {envs_bad[:16]}")


fails = 0
t0 = time.time()
one = rollout(make_env(n, "starpilot"), acts, keep_frames=True)
joint_one = rollout(make_env(n, ",".join(names)), acts)
for it in range(loops):
    for name, f in (
        ("single again", lambda: compare(one, rollout(make_env(n, "starpilot"), acts, keep_frames=True), "single")),
        ("G=2", lambda: compare(one, rollout(make_env(n, "starpilot", extra_options={"num_devices": 2}), acts, keep_frames=True), "G=2")),
        ("G=4", lambda: compare(one, rollout(make_env(n, "starpilot", extra_options={"num_devices": 4}), acts, keep_frames=True), "G=4")),
        ("joint again", lambda: compare(joint_one, rollout(make_env(n, ",".join(names)), acts), "joint")),
        ("joint x 2 shards", lambda: compare(joint_one, rollout(make_env(n, ",".join(names), extra_options={"num_devices": 2}), acts), "joint x2")),
    ):
        try:
            f()
        except Exception as ex:
            fails += 1
            print(f"[{tag}] loop {it} FAIL {name}: {ex}", flush=True)
    try:
        env = make_env(n, ",".join(names), extra_options={"num_devices": 2})
        sts = env.get_state()
        env2 = make_env(n, ",".join(names), rand_seed=5, extra_options={"num_devices": 2})
        env2.set_state(sts)
        got = env2.get_state()
        diff = [e for e in range(n) if got[e] != sts[e]]
        if diff:
            raise AssertionError(f"states differ for envs {diff[:8]} ({len(diff)})")
        env.close()
        env2.close()
    except Exception as ex:
        fails += 1
        print(f"[{tag}] loop {it} FAIL state io: {ex}", flush=True)
print(f"[{tag}] {loops} loops, {fails} failing sections, {time.time() - t0:.0f} s, HOST_THREADS={os.environ.get('PROCGEN_AMD_HOST_THREADS', 'default')}", flush=True)
sys.exit(min(fails, 100))
