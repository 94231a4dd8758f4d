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
