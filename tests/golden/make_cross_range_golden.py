"""
Generates tests/golden/cross_range_state.npz from the COMPILED REFERENCE: a state saved by a handle with num_levels=3, start_level=100
is restored into a handle made with num_levels=0 (and another rand_seed).  The reference adopts the serialized level seed range per env
(src/game.cpp:247-248), so the restored envs go on drawing their levels from [100, 103).  Per game: the two states, the actions
(with forced resets, action -1), and rew / first / level_seed / frame CRC32 of the continuation.

    python tests/golden/make_cross_range_golden.py
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

GAMES = ["coinrun", "maze", "bigfish"]
T0, T1 = 20, 90


def actions(n, steps, seed):
    rng = np.random.RandomState(seed)
    a = rng.randint(0, 15, size=(steps, n)).astype(np.int32)
    a[rng.rand(steps, n) < 0.08] = -1
    return a


def main():
    out = {}
    for game in GAMES:
        a = ref_env.make_ref_env(2, game, rand_seed=3, num_levels=3, start_level=100)
        acts = actions(2, T0 + T1, 9)
        a.observe()
        for t in range(T0):
            a.act(acts[t])
        a.observe()
        states = a.get_state()
        b = ref_env.make_ref_env(2, game, rand_seed=77, num_levels=0)
        b.observe()
        b.set_state(states)
        rec = {k: [] for k in ("rew", "first", "level_seed", "crc")}
        for t in range(T0, T0 + T1 + 1):
            rew, ob, first = b.observe()
            rec["rew"].append(rew.copy())
            rec["first"].append(first.astype(np.uint8))
            rec["level_seed"].append(b.info_arrays()["level_seed"].copy())
            rec["crc"].append(np.array([zlib.crc32(ob["rgb"][e].tobytes()) for e in range(2)], dtype=np.uint32))
            if t < T0 + T1:
                b.act(acts[t])
        end = b.get_state()
        assert set(np.array(rec["level_seed"]).ravel().tolist()) <= {100, 101, 102}, "the restored envs must stay in the saved range"
        for e in range(2):
            out[f"{game}/state{e}"] = np.frombuffer(states[e], dtype=np.uint8).copy()
            out[f"{game}/end{e}"] = np.frombuffer(end[e], dtype=np.uint8).copy()
        out[f"{game}/actions"] = acts[T0:]
        for k, v in rec.items():
            out[f"{game}/{k}"] = np.array(v)
        print(game, "levels seen:", sorted(set(np.array(rec["level_seed"]).ravel().tolist())), "episodes:", int(np.array(rec["first"]).sum()))
        a.close()
        b.close()
    np.savez_compressed(os.path.join(HERE, "cross_range_state.npz"), **out)


if __name__ == "__main__":
    main()
