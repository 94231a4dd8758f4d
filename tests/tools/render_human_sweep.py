"""
TEST TOOL (build container: needs the compiled reference oracle/_ref): the render_human frames (procgen_amd/csrc/pg_human.h, run by the
wave emulation tests/emu) against the COMPILED REFERENCE over every (game, distribution_mode, center_agent) the reference accepts --
the committed fixture tests/golden/render_human.npz holds the default mode only.

    python tests/tools/render_human_sweep.py [envs] [steps] [every] [seed] [game ...]

Prints one line per configuration: frames compared, frames that differ, worst channel difference.
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests", "emu")):
    sys.path.insert(0, p)

ALL = ["bigfish", "bossfight", "caveflyer", "chaser", "climber", "coinrun", "dodgeball", "fruitbot", "heist", "jumper", "leaper", "maze", "miner", "ninja", "plunder", "starpilot"]
EXT = {"chaser", "dodgeball", "leaper", "starpilot"}
MEM = {"caveflyer", "dodgeball", "heist", "jumper", "maze", "miner"}
MODES = {"easy": 0, "hard": 1, "extreme": 2, "memory": 10}


def configs(games):
    for game in games:
        for mode in ["easy", "hard"] + (["extreme"] if game in EXT else []) + (["memory"] if game in MEM else []):
            for center in (True, False):
                yield game, mode, center


def run(game, mode, center, n, steps, every, seed):
    import emu_harness
    import ref_env

    ref = ref_env.make_ref_env(n, game, rand_seed=seed, render_mode="rgb_array", distribution_mode=mode, center_agent=center)
    emu = emu_harness.EmuEnv(n, game, rand_seed=seed, distribution_mode=MODES[mode], center_agent=center)
    rng = np.random.RandomState(seed)
    tot = bad = worst = npx = 0
    for t in range(steps + 1):
        ref.observe()
        emu.observe()
        if t % every == 0 or t == steps:
            hr = ref.info_arrays()["rgb"]
            for e in range(n):
                d = np.abs(hr[e].astype(int) - emu.render_human(e).astype(int))
                tot += 1
                if d.max() > 0:
                    bad += 1
                    worst = max(worst, int(d.max()))
                    npx += int(np.count_nonzero(d.max(axis=2)))
        ac = rng.randint(0, 15, size=(n,), dtype=np.int32)
        ref.act(ac)
        emu.act(ac)
    ref.close()
    emu.close()
    return tot, bad, worst, npx


def main():
    a = sys.argv[1:]
    n = int(a[0]) if len(a) > 0 else 2
    steps = int(a[1]) if len(a) > 1 else 120
    every = int(a[2]) if len(a) > 2 else 15
    seed = int(a[3]) if len(a) > 3 else 5
    games = a[4:] or ALL
    total_bad = 0
    for game, mode, center in configs(games):
        tot, bad, worst, npx = run(game, mode, center, n, steps, every, seed)
        total_bad += bad
        print(f"{game:10s} {mode:8s} center_agent={int(center)}: {tot} frames, {bad} differ ({npx} pixels), worst {worst}", flush=True)
    print("frames that differ:", total_bad)


if __name__ == "__main__":
    main()
