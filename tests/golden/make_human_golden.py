"""
Generates tests/golden/render_human.npz from the COMPILED REFERENCE (oracle/_ref/libenv.so, the unmodified sources under
/root/reference built by `make -C oracle ref`): the 512 x 512 x 3 info "rgb" frames of render_mode="rgb_array"
(reference procgen/env.py:100-105, src/vecgame.cpp:270-282,363-376).  Run in the build container:

    python tests/golden/make_human_golden.py

Per game: 2 envs, rand_seed 7, actions
RandomState(1).randint(0, 15), frames taken at steps 0, 17 and 40:
  <game>/crc      [3][2] CRC32 of each frame's bytes
  <game>/actions  [40][2]
  <game>/state    get_state bytes of env 0 after step 40 (the camera scalars in it are those of the 512-pixel frame)
  frames/<game>   one full frame (step 17, env 0) for coinrun, starpilot, fruitbot and jumper, so that a mismatch can be looked at
Extra option sets for coinrun (center_agent off + paint_vel_info, monochrome assets without backgrounds) under <game>@<k>.
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))

import ref_env  # noqa: E402

GAMES = ["bigfish", "bossfight", "caveflyer", "chaser", "climber", "coinrun", "dodgeball", "fruitbot", "heist", "jumper", "leaper", "maze", "miner", "ninja", "plunder", "starpilot"]
STEPS = [0, 17, 40]
OPTION_SETS = {  # key -> (game, kwargs)
    "coinrun@1": ("coinrun", dict(center_agent=False, paint_vel_info=True)),
    "maze@1": ("maze", dict(use_monochrome_assets=True, use_backgrounds=False)),
    "dodgeball@1": ("dodgeball", dict(distribution_mode="memory", restrict_themes=True)),
    "jumper@1": ("jumper", dict(distribution_mode="easy")),
    "jumper@2": ("jumper", dict(distribution_mode="memory", center_agent=False)),
    # with generated assets (no state: BasicAbstractGame::serialize asserts !use_generated_assets)
    "coinrun@gen": ("coinrun", dict(use_generated_assets=True)),
    "starpilot@gen": ("starpilot", dict(use_generated_assets=True)),
    "fruitbot@gen": ("fruitbot", dict(use_generated_assets=True)),
}
FULL = {"coinrun", "starpilot", "fruitbot", "jumper"}


def run(game, kwargs):
    n = 2
    env = ref_env.make_ref_env(n, game, rand_seed=7, render_mode="rgb_array", **kwargs)
    rng = np.random.RandomState(1)
    crc, acts, full = [], [], None
    for t in range(STEPS[-1] + 1):
        env.observe()
        if t in STEPS:
            rgb = env.info_arrays()["rgb"]
            assert rgb.shape == (n, 512, 512, 3)
            crc.append([zlib.crc32(rgb[e].tobytes()) for e in range(n)])
            if t == 17:
                full = rgb[0].copy()
        if t == STEPS[-1]:
            break
        ac = rng.randint(0, 15, size=(n,), dtype=np.int32)
        acts.append(ac)
        env.act(ac)
    state = np.zeros(0, np.uint8) if kwargs.get("use_generated_assets") else np.frombuffer(env.get_state()[0], dtype=np.uint8).copy()
    env.close()
    return np.array(crc, dtype=np.uint32), np.array(acts, dtype=np.int32), state, full


def main():
    out = {}
    jobs = [(g, g, {}) for g in GAMES] + [(k, g, kw) for k, (g, kw) in OPTION_SETS.items()]
    for key, game, kw in jobs:
        crc, acts, state, full = run(game, kw)
        out[f"{key}/crc"] = crc
        out[f"{key}/actions"] = acts
        out[f"{key}/state"] = state
        if key in FULL:
            out[f"frames/{key}"] = full
        print(key, crc.tolist(), len(state))
    np.savez_compressed(os.path.join(HERE, "render_human.npz"), **out)


if __name__ == "__main__":
    main()
