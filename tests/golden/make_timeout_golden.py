"""
Generates tests/golden/timeout_states.npz from the COMPILED REFERENCE (oracle/_ref, unmodified sources):
the `cur_time >= timeout` path of Game::step (reference src/game.cpp:134) for the timeout classes random or idle play
never reaches (bossfight / plunder 4000 steps, bigfish 6000: the agent dies long before).  A state saved by the reference
gets its serialized `cur_time` moved to a few steps before the game's timeout, is restored into the reference, and the
next steps (random actions) are recorded: rew, first, info and frame CRCs.  The GPU test restores the same patched bytes
through the C ABI's set_state and must reproduce the recording.

    python tests/golden/make_timeout_golden.py        (needs /root/reference and oracle/_ref; run in the build container)
"""
import os
import struct
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests"), os.path.join(REPO, "tests", "tools")):
    sys.path.insert(0, p)
import ref_env  # noqa: E402
import state_parse  # noqa: E402

GAMES = {"bossfight": 4000, "plunder": 4000, "bigfish": 6000, "coinrun": 1000, "leaper": 500}
N, WARM, STEPS = 4, 60, 70


def main():
    out = {}
    for game, timeout in GAMES.items():
        env = ref_env.make_ref_env(N, game, rand_seed=23)
        rng = np.random.RandomState(5)
        for _ in range(WARM):
            env.act(rng.randint(0, 15, size=(N,), dtype=np.int32))
        env.observe()
        states = []
        for e, st in enumerate(env.get_state()):
            p = state_parse.parse_state(st)
            assert p["timeout"] == timeout, (game, p["timeout"])
            b = bytearray(st)
            struct.pack_into("<i", b, p["offset_of_cur_time"], timeout - 12 - 9 * e)
            states.append(bytes(b))
        env.set_state(states)
        rec = {k: [] for k in ("rew", "first", "prev_level_seed", "prev_level_complete", "level_seed", "crc")}
        acts = []
        for t in range(STEPS + 1):
            rew, ob, first = env.observe()
            info = env.info_arrays()
            rec["rew"].append(rew.copy())
            rec["first"].append(np.asarray(first).astype(np.uint8))
            for k in ("prev_level_seed", "prev_level_complete", "level_seed"):
                rec[k].append(info[k].copy())
            rec["crc"].append(np.array([zlib.crc32(ob["rgb"][e].tobytes()) for e in range(N)], dtype=np.uint32))
            if t < STEPS:
                a = rng.randint(0, 15, size=(N,), dtype=np.int32)
                acts.append(a)
                env.act(a)
        env.close()
        first = np.array(rec["first"])
        hit = [int(np.argmax(first[1:, e])) + 1 for e in range(N)]
        assert all(first[1:, e].any() for e in range(N)), game
        print(game, "first episode end per env at recorded step", hit, "(expected by timeout:", [12 + 9 * e for e in range(N)], ")")
        for k, v in rec.items():
            out[f"{game}/{k}"] = np.array(v)
        out[f"{game}/actions"] = np.array(acts)
        for e, st in enumerate(states):
            out[f"{game}/state{e}"] = np.frombuffer(st, dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "timeout_states.npz"), **out)


if __name__ == "__main__":
    main()
