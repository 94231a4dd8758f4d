"""
Generates the committed golden fixtures from the COMPILED REFERENCE (oracle/_ref/libenv.so, built by
`make -C oracle ref` from the unmodified sources under /root/reference).  Run in the build container:

    python tests/golden/make_golden.py

Fixtures (npz, small):
  <game>_rollout.npz : the reference's own determinism protocol (reference procgen/env_test.py:33-52:
                       rand_seed=23, actions = RandomState(0).randint(0, 15, (num,), int32) per step) extended to
                       N envs x T steps: actions, rew, first, prev_level_seed, prev_level_complete, level_seed,
                       per-frame CRC32 of the RGB888 frame, full frames every `frame_every` steps, and the entity
                       table / grid / key scalars parsed from get_state at a few checkpoints.
  mode_matrix.npz    : (`make_golden.py modes`) every accepted (game, distribution_mode) pair besides the default:
                       rew / first / level_seed / frame CRC32 of 6 envs x 100 steps.
  option_matrix.npz  : (`make_golden.py options`) 7 option sets x 16 games (rand_seed=7, actions RandomState(1)): the same
                       four arrays for 6 envs x 60 steps.
  <game>_seeding.npz : reference procgen/env_test.py:7-30 (num_levels=1, start_level in {0,1}): first frames after
                       one step of action 0.
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, os.path.join(REPO, "tests", "tools"))

import ref_env  # noqa: E402
import state_parse  # noqa: E402


def rollout(game, n, t_steps, frame_every, state_at):
    env = ref_env.make_ref_env(n, game, rand_seed=23)
    rng = np.random.RandomState(0)
    out = {k: [] for k in ("actions", "rew", "first", "prev_level_seed", "prev_level_complete", "level_seed", "crc")}
    frames, frame_t = [], []
    states = {}
    for t in range(t_steps + 1):
        rew, ob, first = env.observe()
        info = env.info_arrays()
        out["rew"].append(rew.copy())
        out["first"].append(first.astype(np.uint8))
        for k in ("prev_level_seed", "prev_level_complete", "level_seed"):
            out[k].append(info[k].copy())
        out["crc"].append(np.array([zlib.crc32(ob["rgb"][e].tobytes()) for e in range(n)], dtype=np.uint32))
        if t % frame_every == 0:
            frames.append(ob["rgb"][: min(n, 4)].copy())
            frame_t.append(t)
        if t in state_at:
            raw = env.get_state()
            sts = [state_parse.parse_state(s) for s in raw]
            states[t] = sts
            for e in range(min(n, 2)):
                out.setdefault("_raw", {})[(t, e)] = np.frombuffer(raw[e], dtype=np.uint8).copy()
        ac = rng.randint(0, 15, size=(n,), dtype=np.int32)
        out["actions"].append(ac)
        env.act(ac)
    raw_states = out.pop("_raw", {})
    res = {k: np.array(v) for k, v in out.items()}
    for (t, e), b in raw_states.items():
        res[f"state{t}_e{e}_bytes"] = b  # the reference's get_state byte stream, verbatim
    res["frames"] = np.array(frames)
    res["frame_t"] = np.array(frame_t)
    for t, sts in states.items():
        for e, st in enumerate(sts[: min(n, 4)]):
            res[f"state{t}_e{e}_entities"] = state_parse.entities_as_words(st)
            res[f"state{t}_e{e}_grid"] = st["grid"].astype(np.int16)
            res[f"state{t}_e{e}_scalars"] = np.array(
                [st["cur_time"], st["step_rand_int"], st["background_index"], np.float32(st["bg_pct_x"]).view(np.int32),
                 st["rand_gen"]["idx"], st["current_level_seed"], st["last_move_action"]], dtype=np.int64)
    env.close()
    return res


def seeding(game):
    res = {}
    for lvl in (0, 1):
        env = ref_env.make_ref_env(1, game, num_levels=1, start_level=lvl, rand_seed=5)
        env.act(np.zeros(1, np.int32))
        _, ob, _ = env.observe()
        res[f"level{lvl}"] = ob["rgb"][0].copy()
        env.close()
    return res


# distribution modes other than the default (reference src/game.cpp:55-62 and each game's choose_world_dim / game_reset):
# every (game, mode) pair the reference accepts: 16 easy + 4 extreme + 6 memory (hard is the default the rollouts cover)
MODE_MATRIX = ([(g, "easy") for g in ("bigfish", "bossfight", "caveflyer", "chaser", "climber", "coinrun", "dodgeball", "fruitbot", "heist", "jumper", "leaper",
                                      "maze", "miner", "ninja", "plunder", "starpilot")]
               + [(g, "extreme") for g in ("chaser", "dodgeball", "leaper", "starpilot")]
               + [(g, "memory") for g in ("caveflyer", "dodgeball", "heist", "jumper", "maze", "miner")])


def mode_matrix(n=6, t_steps=100):
    """<game>/<mode>/{rew, first, level_seed, crc}: n envs x t_steps of the compiled reference in that mode; actions as above."""
    res = {}
    for game, mode in MODE_MATRIX:
        env = ref_env.make_ref_env(n, game, rand_seed=23, distribution_mode=mode)
        rng = np.random.RandomState(0)
        out = {k: [] for k in ("rew", "first", "level_seed", "crc")}
        for t in range(t_steps + 1):
            rew, ob, first = env.observe()
            out["rew"].append(rew.copy())
            out["first"].append(first.astype(np.uint8))
            out["level_seed"].append(env.info_arrays()["level_seed"].copy())
            out["crc"].append(np.array([zlib.crc32(ob["rgb"][e].tobytes()) for e in range(n)], dtype=np.uint32))
            env.act(rng.randint(0, 15, size=(n,), dtype=np.int32))
        env.close()
        for k, v in out.items():
            res[f"{game}/{mode}/{k}"] = np.array(v)
    return res


# the option surface of reference src/game.cpp:42-75 that changes frames or level selection, on every game
OPTION_SETS = {
    "no_backgrounds": dict(use_backgrounds=False),
    "no_center_agent": dict(center_agent=False),
    "restrict_themes": dict(restrict_themes=True),
    "two_levels": dict(num_levels=2, start_level=5),
    "sequential_levels": dict(use_sequential_levels=True, num_levels=3),
    "monochrome": dict(use_monochrome_assets=True),
    "vel_info": dict(paint_vel_info=True),
}
ALL_GAMES = ["bigfish", "bossfight", "caveflyer", "chaser", "climber", "coinrun", "dodgeball", "fruitbot", "heist", "jumper", "leaper", "maze", "miner", "ninja",
             "plunder", "starpilot"]


def option_matrix(n=6, t_steps=60):
    """<game>/<option set>/{rew, first, level_seed, crc} for all 16 x 7 pairs (jumper without center_agent draws its compass
    on a non-integer rect: Qt's path engine, pg_qtpath.h)."""
    res = {}
    for game in ALL_GAMES:
        for name, kw in OPTION_SETS.items():
            env = ref_env.make_ref_env(n, game, rand_seed=7, **kw)
            rng = np.random.RandomState(1)
            out = {k: [] for k in ("rew", "first", "level_seed", "crc")}
            for t in range(t_steps + 1):
                rew, ob, first = env.observe()
                out["rew"].append(rew.copy())
                out["first"].append(first.astype(np.uint8))
                out["level_seed"].append(env.info_arrays()["level_seed"].copy())
                out["crc"].append(np.array([zlib.crc32(ob["rgb"][e].tobytes()) for e in range(n)], dtype=np.uint32))
                env.act(rng.randint(0, 15, size=(n,), dtype=np.int32))
            env.close()
            for k, v in out.items():
                res[f"{game}/{name}/{k}"] = np.array(v)
    return res


def generated_assets(n=4, t_steps=60):
    """use_generated_assets=True (reference src/assetgen.cpp, BAG:79-123,769-773): <game>/{rew, first, level_seed, crc} for all 16 games
    plus, for four of them, every frame of env 0 (so that a failing CRC can be looked at)."""
    res = {}
    for game in ALL_GAMES:
        env = ref_env.make_ref_env(n, game, rand_seed=19, use_generated_assets=True)
        rng = np.random.RandomState(2)
        out = {k: [] for k in ("rew", "first", "level_seed", "crc")}
        frames = []
        for t in range(t_steps + 1):
            rew, ob, first = env.observe()
            out["rew"].append(rew.copy())
            out["first"].append(first.astype(np.uint8))
            out["level_seed"].append(env.info_arrays()["level_seed"].copy())
            out["crc"].append(np.array([zlib.crc32(ob["rgb"][e].tobytes()) for e in range(n)], dtype=np.uint32))
            frames.append(ob["rgb"][0].copy())
            env.act(rng.randint(0, 15, size=(n,), dtype=np.int32))
        env.close()
        for k, v in out.items():
            res[f"{game}/{k}"] = np.array(v)
        if game in ("coinrun", "starpilot", "fruitbot", "caveflyer"):
            res[f"{game}/frames0"] = np.array(frames)
    return res


if __name__ == "__main__":
    if sys.argv[1:] == ["generated"]:
        np.savez_compressed(os.path.join(HERE, "generated_assets.npz"), **generated_assets())
        print("generated-assets fixture done")
        sys.exit(0)
    if sys.argv[1:] == ["options"]:
        np.savez_compressed(os.path.join(HERE, "option_matrix.npz"), **option_matrix())
        print("option matrix done")
        sys.exit(0)
    if sys.argv[1:] == ["modes"]:
        np.savez_compressed(os.path.join(HERE, "mode_matrix.npz"), **mode_matrix())
        print("mode matrix done")
        sys.exit(0)
    games = sys.argv[1:] or ["coinrun"]
    for game in games:
        r = rollout(game, n=16, t_steps=512, frame_every=64, state_at=(0, 100, 300, 512))
        np.savez_compressed(os.path.join(HERE, f"{game}_rollout.npz"), **r)
        np.savez_compressed(os.path.join(HERE, f"{game}_seeding.npz"), **seeding(game))
        print(game, "done", {k: v.shape for k, v in r.items() if not k.startswith("state")})
