"""
get_state / set_state (reference src/vecgame.cpp:437-457): the product's wire-format code (procgen_amd/csrc/state_io.cpp)
against the reference's own byte streams recorded in tests/golden (CPU, on the emulated kernels), following the protocol of
reference procgen/state_test.py:71-124.
"""
import os

import numpy as np
import pytest

import emu_harness
from helpers import rollout

GAMES = ["coinrun", "bigfish", "maze", "climber", "miner", "starpilot", "fruitbot", "leaper", "plunder", "heist", "ninja", "dodgeball", "bossfight", "chaser", "caveflyer", "jumper"]


@pytest.mark.parametrize("game", GAMES)
def test_get_state_bytes_identical_to_reference(golden_dir, game):
    g = np.load(os.path.join(golden_dir, f"{game}_rollout.npz"))
    n = g["actions"].shape[1]
    env = emu_harness.EmuEnv(n, game, rand_seed=23)
    t = 0
    for cp in (0, 100, 300, 512):
        while t < cp:
            env.act(g["actions"][t])
            t += 1
        sts = env.get_state()
        for e in range(2):
            assert sts[e] == bytes(g[f"state{cp}_e{e}_bytes"]), f"state bytes of env {e} at step {cp}"


@pytest.mark.parametrize("game", GAMES)
def test_set_state_of_reference_bytes_resumes_the_reference_rollout(golden_dir, game):
    """Restore the REFERENCE's state (step 100, envs 0-1) into a fresh env with another rand_seed: the tail of the
    rollout must reproduce the reference's (state_test.py:103-124 'restore at midpoint')."""
    g = np.load(os.path.join(golden_dir, f"{game}_rollout.npz"))
    env = emu_harness.EmuEnv(2, game, rand_seed=777)
    env.set_state([bytes(g["state100_e0_bytes"]), bytes(g["state100_e1_bytes"])])
    acts = [a[:2] for a in g["actions"][100:300]]
    got = rollout(env, acts)
    for k in ("rew", "first", "prev_level_seed", "prev_level_complete", "level_seed", "crc"):
        assert np.array_equal(got[k], g[k][100:301, :2]), k


def test_get_state_is_non_perturbing_and_roundtrips():
    """state_test.py:88-101: saving and restoring every step is transparent."""
    acts = [np.random.RandomState(3).randint(0, 15, size=(3,), dtype=np.int32) for _ in range(40)]
    a = rollout(emu_harness.EmuEnv(3, "coinrun", rand_seed=9), acts)
    env = emu_harness.EmuEnv(3, "coinrun", rand_seed=9)
    out = []
    for t in range(len(acts) + 1):
        st = env.get_state()
        env.set_state(st)
        assert env.get_state() == st
        _, ob, _ = env.observe()
        out.append(ob["rgb"].copy())
        if t < len(acts):
            env.act(acts[t])
    import zlib

    crc = np.array([[zlib.crc32(f[e].tobytes()) for e in range(3)] for f in out], dtype=np.uint32)
    assert np.array_equal(crc, a["crc"])


def _replay_cross_range(gold, game, env):
    import zlib

    env.observe()
    env.set_state([gold[f"{game}/state0"].tobytes(), gold[f"{game}/state1"].tobytes()])
    acts = gold[f"{game}/actions"]
    for t in range(len(acts) + 1):
        rew, ob, first = env.observe()
        assert np.array_equal(rew, gold[f"{game}/rew"][t]) and np.array_equal(first.astype(np.uint8), gold[f"{game}/first"][t]), f"{game} step {t}"
        assert np.array_equal(env.info_arrays()["level_seed"], gold[f"{game}/level_seed"][t]), f"{game} step {t}: level seeds"
        assert [zlib.crc32(ob["rgb"][e].tobytes()) for e in range(2)] == list(gold[f"{game}/crc"][t]), f"{game} step {t}: frames"
        if t < len(acts):
            env.act(acts[t])
    assert env.get_state() == [gold[f"{game}/end0"].tobytes(), gold[f"{game}/end1"].tobytes()]


@pytest.mark.parametrize("game", ["coinrun", "maze", "bigfish"])
def test_restored_envs_keep_the_level_seed_range_they_were_saved_under(golden_dir, game):
    """reference src/game.cpp:247-248: deserialize adopts level_seed_low / high per env.  tests/golden/cross_range_state.npz (compiled
    reference): states saved under num_levels=3, start_level=100 restored into a num_levels=0 handle keep drawing levels 100..102."""
    gold = np.load(os.path.join(golden_dir, "cross_range_state.npz"))
    env = emu_harness.EmuEnv(2, game, rand_seed=77, num_levels=0)
    _replay_cross_range(gold, game, env)
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("game", ["coinrun", "maze", "bigfish"])
def test_gpu_restored_envs_keep_the_level_seed_range_they_were_saved_under(golden_dir, game):
    from procgen_amd import ProcgenGym3Env

    gold = np.load(os.path.join(golden_dir, "cross_range_state.npz"))
    env = ProcgenGym3Env(2, game, rand_seed=77, num_levels=0)
    _replay_cross_range(gold, game, env)
    env.close()


# ---- options adopted per env (reference src/game.cpp:233-246) --------------------------------------------------------------------------
def _cross_option_cases():
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_cross_option_golden as M

    return M.CASES


CROSS_OPTION_CASES = _cross_option_cases()


def _replay_cross_option(gold, name, env):
    import zlib

    env.observe()
    env.set_state([gold[f"{name}/state0"].tobytes(), gold[f"{name}/state1"].tobytes()])
    acts = gold[f"{name}/actions"]
    for t in range(len(acts) + 1):
        rew, ob, first = env.observe()
        assert np.array_equal(rew, gold[f"{name}/rew"][t]) and np.array_equal(first.astype(np.uint8), gold[f"{name}/first"][t]), f"{name} step {t}"
        assert np.array_equal(env.info_arrays()["level_seed"], gold[f"{name}/level_seed"][t]), f"{name} step {t}: level seeds"
        assert [zlib.crc32(ob["rgb"][e].tobytes()) for e in range(2)] == list(gold[f"{name}/crc"][t]), f"{name} step {t}: frames"
        if t < len(acts):
            env.act(acts[t])
    assert env.get_state() == [gold[f"{name}/end0"].tobytes(), gold[f"{name}/end1"].tobytes()]


@pytest.mark.parametrize("name", sorted(CROSS_OPTION_CASES))
def test_restored_envs_keep_the_game_options_they_were_saved_under(golden_dir, name):
    """reference src/game.cpp:233-246: deserialize adopts the serialized options per env.  tests/golden/cross_option_state.npz (compiled
    reference): states saved under one option set, restored into a handle made with another, continue -- steps, forced resets, frames,
    end states -- exactly as in the reference.  Here: the kernel sources in the CPU emulation."""
    game, _saved, made = CROSS_OPTION_CASES[name]
    gold = np.load(os.path.join(golden_dir, "cross_option_state.npz"))
    made = dict(made)
    if "distribution_mode" in made:  # (the emulation's constructor takes the option's integer value)
        from procgen_amd.env import DISTRIBUTION_MODE_DICT

        made["distribution_mode"] = DISTRIBUTION_MODE_DICT[made["distribution_mode"]]
    env = emu_harness.EmuEnv(2, game, rand_seed=88, **made)
    _replay_cross_option(gold, name, env)
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CROSS_OPTION_CASES))
def test_gpu_restored_envs_keep_the_game_options_they_were_saved_under(golden_dir, name):
    from procgen_amd import ProcgenGym3Env

    game, _saved, made = CROSS_OPTION_CASES[name]
    gold = np.load(os.path.join(golden_dir, "cross_option_state.npz"))
    env = ProcgenGym3Env(2, game, rand_seed=88, **made)
    _replay_cross_option(gold, name, env)
    env.close()


@pytest.mark.gpu
def test_gpu_states_of_another_distribution_mode_are_refused_with_the_reason(golden_dir):
    """what still cannot be adopted per env: caveflyer's memory mode has kernels (and LDS arenas) of its own, and use_generated_assets selects
    the assets and the renderer of a handle.  Every other mode change is adopted (cross_option_state.npz, the */X_into_Y cases)."""
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); from procgen_amd import ProcgenGym3Env; "
            "a = ProcgenGym3Env(1, 'caveflyer', distribution_mode='memory'); a.observe(); st = a.get_state(); "
            "b = ProcgenGym3Env(1, 'caveflyer', distribution_mode='hard'); b.observe(); b.set_state(st)") % repo
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode != 0 and "distribution_mode" in (r.stdout + r.stderr)
    # a chaser state of a LARGER maze than the handle's mode generates: adopted, and ends the run at the env's next reset as the
    # reference does (its per-Game MazeGen keeps the first reset's dimension, chaser.cpp:159-162; grid.h:41 fassert)
    code = ("import sys; sys.path.insert(0, %r); import numpy as np; from procgen_amd import ProcgenGym3Env; "
            "a = ProcgenGym3Env(1, 'chaser', distribution_mode='extreme'); a.observe(); st = a.get_state(); "
            "b = ProcgenGym3Env(1, 'chaser', distribution_mode='hard'); b.observe(); b.set_state(st); b.observe(); print('restored', flush=True); "
            "b.act(np.array([-1], dtype=np.int32)); b.observe(); print('survived', flush=True)") % repo
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode != 0 and "restored" in r.stdout and "survived" not in r.stdout and "device-side check failed" in r.stdout


@pytest.mark.gpu
def test_gpu_states_restored_at_other_indices_come_back_byte_for_byte():
    """reference src/game.cpp:193,253: game_n -- the env's index -- is part of the stream, adopted by deserialize and written back by
    serialize (it only names the env in a warning).  States restored at OTHER indices than they were saved at therefore come back from
    get_state unchanged (checked on the compiled reference: after set_state([st[1], st[0]]) its get_state returns [st[1], st[0]])."""
    from procgen_amd import ProcgenGym3Env

    n = 6
    env = ProcgenGym3Env(n, "coinrun", rand_seed=5)
    rng = np.random.RandomState(2)
    for _ in range(7):
        env.act(rng.randint(0, 15, size=(n,), dtype=np.int32))
    env.observe()
    st = env.get_state()
    assert len(set(st)) == n
    perm = [3, 0, 5, 1, 2, 4]
    env.set_state([st[p] for p in perm])
    assert env.get_state() == [st[p] for p in perm]
    # ... and the restored envs go on exactly like the envs the states came from
    twin = ProcgenGym3Env(n, "coinrun", rand_seed=5)
    rng = np.random.RandomState(2)
    for _ in range(7):
        twin.act(rng.randint(0, 15, size=(n,), dtype=np.int32))
    for _ in range(12):
        a = rng.randint(0, 15, size=(n,), dtype=np.int32)
        twin.act(a)
        env.act(a[perm])
        _, ob_t, _ = twin.observe()
        _, ob_e, _ = env.observe()
        assert np.array_equal(ob_e["rgb"], ob_t["rgb"][perm])
    env.close()
    twin.close()
