"""
Generates tests/golden/human_midstate.npz from the COMPILED REFERENCE (oracle/_ref/libenv.so): get_state called BETWEEN libenv_act and
libenv_observe on a render_mode="rgb_array" handle.  The stepping thread has drawn the 64-pixel observation only; the 512-pixel info
frames are drawn by VecGame::observe (reference src/vecgame.cpp:363-376), so the camera scalars in these bytes are the 64-pixel
frame's -- unlike the state taken after an observe (tests/golden/render_human.npz).  Run in the build container:

    python tests/golden/make_human_midstate_golden.py

Per game: 2 envs, rand_seed 7, actions RandomState(1).randint(0, 15); 10 x (act, observe), then act and get_state:
  <game>/actions [11][2], <game>/mid_state (env 0, between act and observe), <game>/after_state (env 0, after the observe that follows)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))

import ref_env  # noqa: E402

GAMES = ["coinrun", "jumper", "bigfish", "maze"]


def main():
    out = {}
    for game in GAMES:
        n = 2
        env = ref_env.make_ref_env(n, game, rand_seed=7, render_mode="rgb_array")
        rng = np.random.RandomState(1)
        acts = rng.randint(0, 15, size=(11, n), dtype=np.int32)
        env.observe()
        for t in range(10):
            env.act(acts[t])
            env.observe()
        env.act(acts[10])
        mid = env.get_state()[0]
        env.observe()
        after = env.get_state()[0]
        env.close()
        assert mid != after, "the two states should differ in the camera scalars"
        out[f"{game}/actions"] = acts
        out[f"{game}/mid_state"] = np.frombuffer(mid, dtype=np.uint8).copy()
        out[f"{game}/after_state"] = np.frombuffer(after, dtype=np.uint8).copy()
        print(game, len(mid), sum(a != b for a, b in zip(mid, after)), "bytes differ")
    np.savez_compressed(os.path.join(HERE, "human_midstate.npz"), **out)


if __name__ == "__main__":
    main()
