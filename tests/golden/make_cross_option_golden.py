"""
Generates tests/golden/cross_option_state.npz from the COMPILED REFERENCE: states saved by a handle made with one set of game options are
restored into a handle made with another.  The reference's Game::deserialize adopts the serialized options per env (src/game.cpp:233-246:
paint_vel_info, use_monochrome_assets, restrict_themes, use_backgrounds, center_agent, debug_mode, use_sequential_levels), so the restored
envs go on playing -- and resetting, and being drawn -- under the options they were saved with, whatever the handle was made with.
Per case: the two states, the actions (with forced resets, action -1), and rew / first / level_seed / frame CRC32 of the continuation and
the end states.

    python tests/golden/make_cross_option_golden.py
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

# restrict_themes is the same on both sides of every case.  The reference masks an image's theme when its per-Game asset cache slot is first
# filled (BAG:79-123 initialize_asset_if_necessary), not when it is drawn, so after a restore that flips the option the images an env shows
# depend on which slots that Game object happened to fill before -- a property of the cache history, not of the state; this library applies
# the adopted option to every draw.
PLAIN = dict(use_backgrounds=False, center_agent=False, paint_vel_info=True)
MONO = dict(use_monochrome_assets=True, use_sequential_levels=True, num_levels=4, start_level=7)
RESTRICT = dict(restrict_themes=True)
EASY, EXTREME, MEMORY = dict(distribution_mode="easy"), dict(distribution_mode="extreme"), dict(distribution_mode="memory")
# case name -> (game, options the states are saved under, options of the handle they are restored into)
CASES = {
    "coinrun/plain_into_default": ("coinrun", PLAIN, {}),
    "coinrun/default_into_plain": ("coinrun", {}, PLAIN),
    "coinrun/mono_into_default": ("coinrun", MONO, {}),
    "maze/plain_into_default": ("maze", PLAIN, {}),
    "jumper/plain_into_default": ("jumper", PLAIN, {}),
    "jumper/default_into_plain": ("jumper", {}, PLAIN),
    "bigfish/plain_into_default": ("bigfish", PLAIN, {}),
    "fruitbot/plain_into_mono": ("fruitbot", PLAIN, MONO),
    "climber/mono_into_plain": ("climber", MONO, PLAIN),
    "ninja/restricted_plain_into_restricted": ("ninja", dict(PLAIN, **RESTRICT), RESTRICT),
    "heist/plain_into_default": ("heist", PLAIN, {}),
    # round 5: the distribution_mode is adopted per env as well (every mode of a game but caveflyer's memory mode runs on the same kernels)
    "coinrun/easy_into_hard": ("coinrun", EASY, {}),
    "coinrun/hard_into_easy": ("coinrun", {}, EASY),
    "bigfish/easy_into_hard": ("bigfish", EASY, {}),
    "bossfight/easy_into_hard": ("bossfight", EASY, {}),
    "caveflyer/easy_into_hard": ("caveflyer", EASY, {}),
    "chaser/hard_into_extreme": ("chaser", {}, EXTREME),
    "climber/easy_into_hard": ("climber", EASY, {}),
    "dodgeball/hard_into_memory": ("dodgeball", {}, MEMORY),
    "dodgeball/memory_into_extreme": ("dodgeball", MEMORY, EXTREME),
    "fruitbot/hard_into_easy": ("fruitbot", {}, EASY),
    "heist/memory_into_easy": ("heist", MEMORY, EASY),
    "jumper/easy_into_hard": ("jumper", EASY, {}),
    "jumper/hard_into_memory": ("jumper", {}, MEMORY),
    "jumper/memory_into_easy": ("jumper", MEMORY, EASY),
    "leaper/extreme_into_easy": ("leaper", EXTREME, EASY),
    "maze/memory_into_hard": ("maze", MEMORY, {}),
    "miner/hard_into_memory": ("miner", {}, MEMORY),
    "ninja/easy_into_hard": ("ninja", EASY, {}),
    "plunder/hard_into_easy": ("plunder", {}, EASY),
    "starpilot/extreme_into_hard": ("starpilot", EXTREME, {}),
}
T0, T1 = 20, 80


def actions(n, steps, seed):
    rng = np.random.RandomState(seed)
    a = rng.randint(0, 15, size=(steps, n)).astype(np.int32)
    a[rng.rand(steps, n) < 0.06] = -1
    return a


def main():
    out = {}
    for name, (game, saved, made) in CASES.items():
        a = ref_env.make_ref_env(2, game, rand_seed=5, **saved)
        acts = actions(2, T0 + T1, 21)
        a.observe()
        for t in range(T0):
            a.act(acts[t])
        a.observe()
        states = a.get_state()
        b = ref_env.make_ref_env(2, game, rand_seed=88, **made)
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
        for e in range(2):
            out[f"{name}/state{e}"] = np.frombuffer(states[e], dtype=np.uint8).copy()
            out[f"{name}/end{e}"] = np.frombuffer(end[e], dtype=np.uint8).copy()
        out[f"{name}/actions"] = acts[T0:]
        for k, v in rec.items():
            out[f"{name}/{k}"] = np.array(v)
        print(name, "episodes:", int(np.array(rec["first"]).sum()), "levels:", sorted(set(np.array(rec["level_seed"]).ravel().tolist()))[:6])
        a.close()
        b.close()
    np.savez_compressed(os.path.join(HERE, "cross_option_state.npz"), **out)


if __name__ == "__main__":
    main()
