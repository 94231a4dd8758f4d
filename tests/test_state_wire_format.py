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
