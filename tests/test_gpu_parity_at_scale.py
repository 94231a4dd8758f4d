"""
GPU (-m gpu): parity at BASELINE.json's sizes and horizons, through the C ABI.

Env n of a vector depends only on (rand_seed, n) (reference src/vecgame.cpp:301-314) and envs never interact, so an
oracle run of the first N' envs -- or of a strided sample of them (oracle_env.OracleEnv(env_offset=, env_stride=)) --
with the matching action columns replays those envs of any larger GPU run.  Bit-exact on
rew / first / info and on every frame (the contract allows +-1 LSB per channel; we hold 0).
"""
import ctypes as C
import os
import zlib

import numpy as np
import pytest

import oracle_env
from helpers import HIP_LIB, action_stream, assert_rollouts_equal, hip_memcpy_dtoh, rollout

pytestmark = pytest.mark.gpu

GAMES = ["coinrun", "bigfish", "maze", "climber", "miner", "starpilot", "fruitbot", "leaper", "plunder", "heist", "ninja", "dodgeball", "bossfight", "chaser", "caveflyer", "jumper"]


def make_env(n, game="coinrun", **kw):
    from procgen_amd import ProcgenGym3Env

    assert os.path.exists(HIP_LIB), "HIP libenv.so missing: run __graft_entry__.build() (there is no fallback path)"
    kw.setdefault("rand_seed", 23)
    return ProcgenGym3Env(n, game, **kw)


class DeviceBuffers(C.Structure):  # include/procgen_amd.h
    _fields_ = [("device_id", C.c_int), ("num_envs", C.c_int), ("stream", C.c_void_p), ("ob", C.c_void_p), ("rew", C.c_void_p),
                ("first", C.c_void_p), ("prev_level_seed", C.c_void_p), ("prev_level_complete", C.c_void_p), ("level_seed", C.c_void_p),
                ("action", C.c_void_p)]


def check_prefix_against_oracle(game, n, m, steps, seed):
    """n envs on the GPU (observations resident in HBM, the first m frames read back through the extension hook every
    step) against an m-env oracle run with the same action columns."""
    env = make_env(n, game, extra_options={"host_observations": False})
    b = DeviceBuffers()
    env._lib.procgen_amd_device_buffers.argtypes = [C.c_void_p, C.POINTER(DeviceBuffers)]
    assert env._lib.procgen_amd_device_buffers(env._handle, C.byref(b)) == 0 and b.num_envs == n
    orc = oracle_env.OracleEnv(m, game, rand_seed=23)
    rng = np.random.RandomState(seed)
    resets = 0
    for t in range(steps + 1):
        rew, _, first = env.observe()
        orew, oob, ofirst = orc.observe()
        assert np.array_equal(rew[:m], orew) and np.array_equal(first[:m], ofirst), f"{game}: rew / first at step {t}"
        for k, v in orc.info_arrays().items():
            assert np.array_equal(env.info_arrays()[k][:m], v), f"{game}: {k} at step {t}"
        frames = hip_memcpy_dtoh(b.ob, m * 12288).reshape(m, 64, 64, 3)
        assert np.array_equal(frames, oob["rgb"]), f"{game}: frame at step {t}, envs {np.nonzero((frames != oob['rgb']).reshape(m, -1).any(axis=1))[0][:8]}"
        if t:
            resets += int(ofirst.sum())
        if t < steps:
            ac = rng.randint(0, 15, size=(n,), dtype=np.int32)
            env.act(ac)
            orc.act(ac[:m])
    env.close()
    orc.close()
    return resets


@pytest.mark.parametrize("game,n", [("coinrun", 65536), ("bigfish", 65536), ("starpilot", 32768)])
def test_baseline_config_sizes_against_a_1024_env_oracle(game, n):
    """BASELINE configs[1], [2] (65536 envs) and the one-GPU share of configs[3] (starpilot, 262144 / 8): the first 1024
    envs x 200 steps equal the oracle's."""
    resets = check_prefix_against_oracle(game, n, 1024, 200, seed=31)
    assert resets > 100  # level generation on the device was exercised many times


def test_sixteen_game_joint_handle_at_its_one_gpu_share():
    """BASELINE configs[4] (all 16 games, 131072 envs over 8 GPUs): the one-GPU share, 16384 envs = 1024 per game, all 16
    games' kernels in flight together.  Env n plays names[n % 16]; a strided sample of 64 envs per game x 200 steps equals
    per-game oracle runs of exactly those envs."""
    K, n, per_game, steps = len(GAMES), 16384, 64, 200
    rng = np.random.RandomState(41)
    acts = [rng.randint(0, 15, size=(n,), dtype=np.int32) for _ in range(steps)]
    joint = rollout(make_env(n, ",".join(GAMES)), acts)
    for k, game in enumerate(GAMES):
        stride = K * (n // K // per_game)  # 64 envs of this game, spread over the whole vector
        idx = np.arange(per_game) * stride + k
        ref = rollout(oracle_env.OracleEnv(per_game, game, rand_seed=23, env_offset=k, env_stride=stride), [a[idx] for a in acts])
        got = {key: joint[key][:, idx] for key in ref}
        assert_rollouts_equal(ref, got, f"joint handle, {game}")


def test_host_landed_observations_of_a_large_handle_arrive_chunk_by_chunk():
    """A handle of >= 32768 envs made with host observations (the gym3 default) steps in four launch chunks and lands every chunk's slice
    of the caller's array on a copy stream of its own, behind that chunk's render kernel (libenv_hip.cpp VecGame::launch).  Every
    frame the caller sees must be the frame in the device buffer, and the first 128 envs must be the oracle's."""
    n, m, steps = 32768, 128, 12
    env = make_env(n, "coinrun")
    b = DeviceBuffers()
    env._lib.procgen_amd_device_buffers.argtypes = [C.c_void_p, C.POINTER(DeviceBuffers)]
    assert env._lib.procgen_amd_device_buffers(env._handle, C.byref(b)) == 0 and b.num_envs == n
    orc = oracle_env.OracleEnv(m, "coinrun", rand_seed=23)
    rng = np.random.RandomState(5)
    for t in range(steps + 1):
        rew, ob, first = env.observe()
        orew, oob, ofirst = orc.observe()
        dev = hip_memcpy_dtoh(b.ob, n * 12288).reshape(n, 64, 64, 3)
        assert np.array_equal(ob["rgb"], dev), f"step {t}: host array differs from the device buffer in envs {np.nonzero((ob['rgb'] != dev).reshape(n, -1).any(axis=1))[0][:8]}"
        assert np.array_equal(ob["rgb"][:m], oob["rgb"]) and np.array_equal(rew[:m], orew) and np.array_equal(first[:m], ofirst), f"step {t}"
        if t < steps:
            ac = rng.randint(0, 15, size=(n,), dtype=np.int32)
            env.act(ac)
            orc.act(ac[:m])
    env.close()
    orc.close()


def test_sixteen_games_over_eight_device_shards(monkeypatch):
    """BASELINE configs[4] in its whole shape on the one GPU there is: ONE handle, num_devices = 8 x 16 games = 128 parts (each a
    VecGame with its own stream and 128 envs), PROCGEN_AMD_FAKE_DEVICES mapping the eight shards onto the visible device(s).  Device g
    owns the global indices [2048 g, 2048 (g + 1)), env n plays names[n % 16] whatever the sharding (reference src/vecgame.cpp:295-314);
    a strided sample of 32 envs per game -- four in every shard -- x 120 steps equals per-game oracle runs of exactly those envs."""
    monkeypatch.setenv("PROCGEN_AMD_FAKE_DEVICES", "1")
    K, G, n, per_game, steps = len(GAMES), 8, 16384, 32, 120
    rng = np.random.RandomState(43)
    acts = [rng.randint(0, 15, size=(n,), dtype=np.int32) for _ in range(steps)]
    env = make_env(n, ",".join(GAMES), extra_options={"num_devices": G})
    env._lib.procgen_amd_part_buffers.restype = C.c_int
    env._lib.procgen_amd_part_buffers.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    assert env._lib.procgen_amd_part_buffers(env._handle, None, 0) == G * K
    joint = rollout(env, acts)
    for k, game in enumerate(GAMES):
        stride = K * (n // K // per_game)  # 32 envs of this game, spread over the whole vector: n // G // stride = 4 per shard
        idx = np.arange(per_game) * stride + k
        assert len(set(idx // (n // G))) == G
        ref = rollout(oracle_env.OracleEnv(per_game, game, rand_seed=23, env_offset=k, env_stride=stride), [a[idx] for a in acts])
        got = {key: joint[key][:, idx] for key in ref}
        assert_rollouts_equal(ref, got, f"16 games x 8 shards, {game}")


def test_full_size_long_horizon_against_a_strided_oracle_sample():
    """BASELINE configs[1] at its full size over a long horizon: coinrun, 65536 envs, 1200 steps -- past the 1000-step timeout, so every
    episode has ended at least once, the envs are desynchronised, trail-heavy envs sit in the tier-1 / tier-2 arenas and the reset rate is
    at its long-run level (the state bench.py's steady_state object measures).  A strided sample of 256 envs (every 256th) is replayed by
    the oracle: rew / first / info at every step, the sample's frames every 100 steps and at the end."""
    game, n, m, steps = "coinrun", 65536, 256, 1200
    stride = n // m
    idx = np.arange(m) * stride
    env = make_env(n, game, extra_options={"host_observations": False})
    b = DeviceBuffers()
    env._lib.procgen_amd_device_buffers.argtypes = [C.c_void_p, C.POINTER(DeviceBuffers)]
    assert env._lib.procgen_amd_device_buffers(env._handle, C.byref(b)) == 0 and b.num_envs == n
    orc = oracle_env.OracleEnv(m, game, rand_seed=23, env_offset=0, env_stride=stride)
    rng = np.random.RandomState(77)
    episodes = 0
    for t in range(steps + 1):
        rew, _, first = env.observe()
        orew, oob, ofirst = orc.observe()
        assert np.array_equal(rew[idx], orew) and np.array_equal(first[idx], ofirst), f"rew / first at step {t}"
        for k, v in orc.info_arrays().items():
            assert np.array_equal(env.info_arrays()[k][idx], v), f"{k} at step {t}"
        if t % 100 == 0 or t == steps:
            for j, e in enumerate(idx):
                frame = hip_memcpy_dtoh(b.ob + int(e) * 12288, 12288).reshape(64, 64, 3)
                assert np.array_equal(frame, oob["rgb"][j]), f"frame of env {e} at step {t}"
        if t:
            episodes += int(ofirst.sum())
        if t < steps:
            ac = rng.randint(0, 15, size=(n,), dtype=np.int32)
            env.act(ac)
            orc.act(ac[idx])
    env.close()
    orc.close()
    assert episodes >= m, f"only {episodes} episode ends in the sample: the horizon did not desynchronise it"


def noop_heavy_actions(n, steps, seed, p_noop=0.97):
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(steps):
        a = rng.randint(0, 15, size=(n,), dtype=np.int32)
        a[rng.rand(n) < p_noop] = 4
        out.append(a)
    return out


@pytest.mark.parametrize("game,timeout", [("coinrun", 1000), ("heist", 1000), ("maze", 500)])
def test_episodes_that_end_by_timeout(game, timeout):
    """Rollouts longer than the game's timeout with mostly no-op actions, so that episodes reach `cur_time >= timeout`
    (reference src/game.cpp:134; timeouts: 1000 default, maze / leaper 500).  The 4000- and 6000-step classes, where idle
    play dies long before, are covered by test_timeout_classes_from_reference_states."""
    n, steps = 8, timeout + 110
    acts = noop_heavy_actions(n, steps, seed=7)
    a = rollout(oracle_env.OracleEnv(n, game, rand_seed=23), acts)
    b = rollout(make_env(n, game), acts)
    assert_rollouts_equal(a, b, f"timeout horizon ({game})")
    assert a["first"][timeout:timeout + 2].any(), "an episode must have run into the timeout"


@pytest.mark.parametrize("game", ["bossfight", "plunder", "bigfish", "coinrun", "leaper"])
def test_timeout_classes_from_reference_states(golden_dir, game):
    """tests/golden/timeout_states.npz (compiled reference, make_timeout_golden.py): reference states whose serialized
    cur_time sits a few steps before the game's timeout (bossfight / plunder 4000, bigfish 6000, coinrun 1000, leaper 500)
    are restored through set_state; rew / first / info / frames of the next 70 steps equal the reference's recording,
    with every env's episode ending exactly when cur_time reaches the timeout."""
    g = np.load(os.path.join(golden_dir, "timeout_states.npz"))
    n = g[f"{game}/actions"].shape[1]
    env = make_env(n, game, rand_seed=777)
    env.set_state([bytes(g[f"{game}/state{e}"]) for e in range(n)])
    got = rollout(env, list(g[f"{game}/actions"]))
    for k in ("rew", "first", "prev_level_seed", "prev_level_complete", "level_seed", "crc"):
        assert np.array_equal(got[k], g[f"{game}/{k}"]), (game, k)
    assert [int(np.argmax(got["first"][1:, e])) + 1 for e in range(n)] == [12 + 9 * e for e in range(n)]


def test_separately_placed_per_env_buffers():
    """libenv_buffers only promises one pointer per env (reference src/vecgame.cpp:30-40): with padding behind every
    env's slice neither the observation nor the action / info arrays are one dense array, so the library takes its
    strided paths (staging buffer for the frames, per-env gathers / scatters)."""
    n, steps = 24, 60
    acts = action_stream(n, steps, seed=6)
    a = rollout(make_env(n), acts, keep_frames=True)
    env = make_env(n, buffer_padding=100)
    assert not env._ob["rgb"].flags["C_CONTIGUOUS"] and env._ac["action"].strides[0] != 4
    b = rollout(env, acts, keep_frames=True)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    for store in env._stores:  # nothing was written into the padding
        assert not store[:, -100:].any()


def test_set_state_leaves_one_list_entry_whatever_the_tier_order():
    """Restoring a many-entity state, then a few-entity state, then the many-entity state again into the same env (no
    step in between) must leave exactly one routing entry for it: the next steps equal the donor's continuation."""
    n, steps = 64, 400
    acts = action_stream(n, steps, seed=8)
    donor = make_env(n, rand_seed=99)
    counts = []
    states = {}
    for t in range(steps):
        donor.act(acts[t])
        if t in (150, 399):
            states[t] = donor.get_state()
    import state_parse

    sizes = {t: [len(state_parse.parse_state(s)["entities"]) for s in states[t]] for t in states}
    big_env = int(np.argmax(sizes[399]))
    small_env = int(np.argmin(sizes[150]))
    assert sizes[399][big_env] > 64 > sizes[150][small_env], (sizes[399][big_env], sizes[150][small_env])
    big, small = states[399][big_env], states[150][small_env]
    # continuation of the donor's env `big_env` from step 400 on, from a single-env handle holding that state
    cont = make_env(1, rand_seed=1)
    cont.set_state([big])
    tail = [np.array([a[big_env]], dtype=np.int32) for a in action_stream(n, 60, seed=9)]
    want = rollout(cont, tail)
    env = make_env(4, rand_seed=5)
    for st in (big, small, big):
        env.call_c_func("set_state", 2, st, len(st))
    got = rollout(env, [np.array([4, 4, a[0], 4], dtype=np.int32) for a in tail])
    for k in want:
        assert np.array_equal(want[k][:, 0], got[k][:, 2]), k


def test_one_handle_sharded_over_devices_equals_the_single_device_handle(monkeypatch):
    """The "num_devices" option (include/procgen_amd.h): one libenv handle, contiguous index ranges
    per device, observations landed in the caller's one (registered) host array, no collective.  Runs with however many
    devices are visible; PROCGEN_AMD_FAKE_DEVICES lets several shards share one GPU, so the sharding logic is exercised on
    a one-GPU box too.  Every output equals the single-device handle's, and get_state / set_state address global indices."""
    import torch

    monkeypatch.setenv("PROCGEN_AMD_FAKE_DEVICES", "1")
    ndev = max(torch.cuda.device_count(), 1)
    n, steps = 96, 90
    acts = action_stream(n, steps, seed=12)
    one = rollout(make_env(n, "starpilot"), acts, keep_frames=True)
    for G in sorted({2, 4, ndev} - {1}):
        if n % G:
            continue
        got = rollout(make_env(n, "starpilot", extra_options={"num_devices": G}), acts, keep_frames=True)
        for k in one:
            assert np.array_equal(one[k], got[k]), (G, k)
    # joint games x device shards: env n plays names[n % K] whatever the sharding
    names = ["coinrun", "bigfish", "maze"]
    joint_one = rollout(make_env(n, ",".join(names)), acts)
    joint_sh = rollout(make_env(n, ",".join(names), extra_options={"num_devices": 2}), acts)
    for k in joint_one:
        assert np.array_equal(joint_one[k], joint_sh[k]), k
    import state_parse

    env = make_env(n, ",".join(names), extra_options={"num_devices": 2})
    sts = env.get_state()
    for e in (0, 47, 48, 95):
        st = state_parse.parse_state(sts[e])
        assert st["game_name"] == names[e % 3] and st["game_n"] == e
    env2 = make_env(n, ",".join(names), rand_seed=5, extra_options={"num_devices": 2})
    env2.set_state(sts)
    assert env2.get_state() == sts
    env.close()
    env2.close()
    # device-resident reader of a multi-part handle (procgen_amd_part_buffers): 2 shards x 3 games = 6 parts whose dense
    # device arrays, read back part by part, are the frames the same handle lands on the host
    class Part(C.Structure):
        _fields_ = [("buffers", DeviceBuffers), ("first_env", C.c_int), ("env_stride", C.c_int), ("game", C.c_char * 128)]

    env = make_env(n, ",".join(names), extra_options={"num_devices": 2})
    for t in range(5):
        env.act(acts[t])
    rew, ob, first = env.observe()
    env._lib.procgen_amd_part_buffers.argtypes = [C.c_void_p, C.POINTER(Part), C.c_int]
    env._lib.procgen_amd_part_buffers.restype = C.c_int
    assert env._lib.procgen_amd_part_buffers(env._handle, None, 0) == 6
    parts = (Part * 6)()
    assert env._lib.procgen_amd_part_buffers(env._handle, parts, 6) == 6
    seen = np.zeros(n, bool)
    for p in parts:
        m = p.buffers.num_envs
        idx = p.first_env + p.env_stride * np.arange(m)
        assert m == n // 6 and p.env_stride == 3 and p.game.decode() == names[p.first_env % 3]
        dev_ob = hip_memcpy_dtoh(p.buffers.ob, m * 64 * 64 * 3).reshape(m, 64, 64, 3)
        dev_rew = hip_memcpy_dtoh(p.buffers.rew, m * 4).view(np.float32)
        assert np.array_equal(dev_ob, ob["rgb"][idx]) and np.array_equal(dev_rew, rew[idx])
        seen[idx] = True
    assert seen.all()
    env.close()
