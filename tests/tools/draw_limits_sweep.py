"""
The renderer's capacity / shape exits (pg_render.h fail(PGE_UNSUPPORTED_DRAW): background tile slots, the compass needle's line
bound, rotated / tiled sprites in a game that never declared them, ...) are not reference semantics: one env hitting one of them
is a fatal exit for the whole handle.  This sweep runs the emulated kernels over every (game, distribution_mode, center_agent)
the reference accepts and counts device error words; the reduced form is a CPU test (tests/test_kernel_logic_emu.py).

    python tests/tools/draw_limits_sweep.py [envs] [steps] [seed]
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "tests", "emu"))
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, REPO)

ALL = ["bigfish", "bossfight", "caveflyer", "chaser", "climber", "coinrun", "dodgeball", "fruitbot", "heist", "jumper", "leaper", "maze", "miner", "ninja", "plunder", "starpilot"]
EXT = {"chaser", "dodgeball", "leaper", "starpilot"}
MEM = {"caveflyer", "dodgeball", "heist", "jumper", "maze", "miner"}
MODES = {"easy": 0, "hard": 1, "extreme": 2, "memory": 10}


def configs():
    for game in ALL:
        for mode in ["easy", "hard"] + (["extreme"] if game in EXT else []) + (["memory"] if game in MEM else []):
            for center in (True, False):
                yield game, mode, center


def run(game, mode, center, envs, steps, seed):
    import emu_harness

    env = emu_harness.EmuEnv(envs, game, rand_seed=seed, distribution_mode=MODES[mode], center_agent=center)
    rng = np.random.RandomState(seed)
    for _ in range(steps):
        env.act(rng.randint(0, 15, size=envs).astype(np.int32))
    env.observe()
    errs = [env.L.emu_error(env.h, e) for e in range(envs)]
    env.close()
    return [e for e in errs if e]


if __name__ == "__main__":
    envs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 101
    total = 0
    for game, mode, center in configs():
        errs = run(game, mode, center, envs, steps, seed)
        total += len(errs)
        print(f"{game:10s} {mode:8s} center_agent={center!s:5s} envs with an error word: {len(errs)} {sorted(set(errs)) if errs else ''}", flush=True)
    print("total", total)
