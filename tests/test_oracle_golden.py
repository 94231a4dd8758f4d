"""
CPU: the plain-C oracle (oracle/procgen_oracle.c) against the committed golden fixtures that were generated from
the COMPILED REFERENCE (tests/golden/make_golden.py).  The reference's own tests hold no value-level vectors
(reference procgen/env_test.py, state_test.py are self-consistency protocols); these fixtures run those protocols
on the reference build and record the values.
"""
import os
import zlib

import numpy as np
import pytest

import oracle_env
from helpers import action_stream, assert_rollouts_equal, check_against_generated_assets_fixture, check_against_option_matrix, rollout


GAMES = ["coinrun", "bigfish", "maze", "climber", "miner", "starpilot", "fruitbot", "leaper", "plunder", "heist", "ninja", "dodgeball", "bossfight", "chaser", "caveflyer", "jumper"]


@pytest.fixture(scope="module", params=GAMES)
def gold(request, golden_dir):
    g = dict(np.load(os.path.join(golden_dir, f"{request.param}_rollout.npz")))
    g["game"] = request.param
    return g


def test_oracle_matches_reference_rollout(gold):
    n = gold["actions"].shape[1]
    env = oracle_env.OracleEnv(n, gold["game"], rand_seed=23)
    actions = list(gold["actions"][:-1])
    got = rollout(env, actions, keep_frames=True)
    ref = {k: gold[k] for k in ("rew", "first", "prev_level_seed", "prev_level_complete", "level_seed", "crc")}
    assert_rollouts_equal(got, ref, "oracle vs compiled reference")
    # full frames at the recorded checkpoints: bit exact (the contract allows +-1 LSB; we hold 0)
    for k, t in enumerate(gold["frame_t"]):
        assert np.array_equal(got["frames"][t][:4], gold["frames"][k]), f"frame at step {t}"
    assert got["first"][1:].sum() > 0, "rollout should contain episode boundaries"
    assert (got["rew"] > 0).sum() >= 0


def test_oracle_entity_tables_match_reference_state(gold):
    """Entity table / grid parsed from the reference's get_state bytes at a few checkpoints."""
    n = gold["actions"].shape[1]
    env = oracle_env.OracleEnv(n, gold["game"], rand_seed=23)
    checkpoints = sorted({int(k.split("_")[0][5:]) for k in gold if k.startswith("state")})
    t = 0
    for cp in checkpoints:
        while t < cp:
            env.act(gold["actions"][t])
            t += 1
        for e in range(4):
            ents = gold[f"state{cp}_e{e}_entities"]
            assert np.array_equal(env.entities(e), ents), f"entities of env {e} at step {cp}"
            assert np.array_equal(env.grid(e), gold[f"state{cp}_e{e}_grid"].astype(np.int32)), f"grid of env {e} at step {cp}"
            sc = gold[f"state{cp}_e{e}_scalars"]
            mine = env.scalars(e)
            assert mine[0] == sc[0] and mine[2] == sc[1] and mine[3] == sc[2] and mine[4] == sc[3], f"scalars of env {e} at step {cp}"


@pytest.mark.parametrize("game", GAMES)
def test_oracle_seeding_protocol(golden_dir, game):
    """reference procgen/env_test.py:7-30: same start_level -> same frame, different level -> different frame."""
    g = np.load(os.path.join(golden_dir, f"{game}_seeding.npz"))
    frames = {}
    for lvl in (0, 1):
        env = oracle_env.OracleEnv(1, game, num_levels=1, start_level=lvl, rand_seed=5)
        env.act(np.zeros(1, np.int32))
        _, ob, _ = env.observe()
        frames[lvl] = ob["rgb"][0].copy()
        assert np.array_equal(frames[lvl], g[f"level{lvl}"])
    assert not np.array_equal(frames[0], frames[1])


def test_oracle_env_independent_of_num_envs():
    """Env n depends only on (rand_seed, n) (reference src/vecgame.cpp:301-314): the basis of sharding."""
    acts = [np.random.RandomState(7).randint(0, 15, size=(6,), dtype=np.int32) for _ in range(40)]
    a = rollout(oracle_env.OracleEnv(6, "coinrun", rand_seed=11), acts)
    b = rollout(oracle_env.OracleEnv(3, "coinrun", rand_seed=11), [x[:3] for x in acts])
    for k in a:
        assert np.array_equal(a[k][:, :3], b[k])


MODE_IDS = {"easy": 0, "hard": 1, "extreme": 2, "memory": 10}


def _mode_pairs(golden_dir):
    g = np.load(os.path.join(golden_dir, "mode_matrix.npz"))
    return g, sorted({tuple(k.split("/")[:2]) for k in g.files})


def test_oracle_matches_reference_in_every_distribution_mode(golden_dir):
    """tests/golden/mode_matrix.npz (compiled reference, `make_golden.py modes`): all accepted (game, mode) pairs besides the default."""
    g, pairs = _mode_pairs(golden_dir)
    assert len(pairs) == 26  # 16 easy + 4 extreme + 6 memory: every pair reference src/game.cpp:55-66 accepts
    for game, mode in pairs:
        n = g[f"{game}/{mode}/rew"].shape[1]
        steps = g[f"{game}/{mode}/rew"].shape[0] - 1
        got = rollout(oracle_env.OracleEnv(n, game, rand_seed=23, distribution_mode=MODE_IDS[mode]), action_stream(n, steps))
        for k in ("rew", "first", "level_seed", "crc"):
            assert np.array_equal(got[k], g[f"{game}/{mode}/{k}"]), (game, mode, k)


def test_oracle_matches_reference_on_the_option_surface(golden_dir):
    """tests/golden/option_matrix.npz (compiled reference): 7 option sets x 16 games.  restrict_themes also masks the theme
    the aspect ratio of an entity comes from (reference src/basic-abstract-game.cpp:82-86,114), which moves bigfish /
    bossfight / fruitbot physics."""
    g = np.load(os.path.join(golden_dir, "option_matrix.npz"))
    pairs = sorted({tuple(k.split("/")[:2]) for k in g.files})
    assert len(pairs) == 16 * 7
    check_against_option_matrix(g, lambda game, n, **kw: oracle_env.OracleEnv(n, game, rand_seed=7, **kw), pairs)


def _qt_ellipse_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "qt_path_ellipses.npz"))
    unpack = lambda k: np.unpackbits(g[k], axis=-1).astype(bool)
    return g["rects"], unpack("both"), unpack("both_brush"), unpack("brush_only"), unpack("pen_only")


def test_oracle_ellipse_restatement_matches_qt_pixels(golden_dir):
    """drawEllipse(QRectF) without antialiasing, as Qt 5.9.7 itself drew it (tests/golden/qt_path_ellipses.npz, written by
    tests/tools/qt_path_probe.py golden with PyQt5 5.9.7): the five jumper compass rects + 1200 random / knife-edge / tiny /
    partly-outside rects; pen + brush, brush alone, pen alone."""
    import ctypes as C

    L = oracle_env.lib()
    L.pgo_test_draw_ellipse.argtypes = [C.c_double] * 4 + [C.c_int, C.c_int, C.c_void_p]
    rects, both_pen, both_brush, brush_only, pen_only = _qt_ellipse_fixture(golden_dir)
    out = np.zeros((64, 64), np.uint8)
    for i, (x, y, w, h) in enumerate(rects):
        L.pgo_test_draw_ellipse(x, y, w, h, 1, 1, out.ctypes.data)
        assert np.array_equal(out == 2, both_pen[i]) and np.array_equal(out == 1, both_brush[i]), ("pen + brush", i, (x, y, w, h))
        L.pgo_test_draw_ellipse(x, y, w, h, 0, 1, out.ctypes.data)
        assert np.array_equal(out == 1, brush_only[i]), ("brush", i, (x, y, w, h))
        L.pgo_test_draw_ellipse(x, y, w, h, 1, 0, out.ctypes.data)
        assert np.array_equal(out == 2, pen_only[i]), ("pen", i, (x, y, w, h))


def test_oracle_with_generated_assets_matches_reference_fixture(golden_dir):
    """use_generated_assets=True on all 16 games: AssetGen's painter (fillRect, path / midpoint ellipses), the per-episode 500 x 500
    backgrounds drawn from rand_gen inside game_reset, and Qt's generic span route for the ARGB32 sprites (rotated ones included)."""
    g = np.load(os.path.join(golden_dir, "generated_assets.npz"))
    check_against_generated_assets_fixture(g, lambda game, n, **kw: oracle_env.OracleEnv(n, game, rand_seed=19, **kw), GAMES)


def test_cffi_binding_script_drives_the_compiled_reference(golden_dir):
    """tests/tools/cffi_replay.py -- gym3's cffi call sequence with the reference's own `c_func_defs` (reference procgen/env.py:128-136) --
    against the reference's own libenv.so (oracle/_ref): the binding the GPU test uses on the HIP library is the one the reference's
    library accepts (libenv_make's by-value options struct, the pointer tables, get_state / set_state through char buffers)."""
    import ref_env
    from test_gpu_parity import cffi_available, run_cffi_replay

    if not ref_env.available():
        pytest.skip("compiled reference not built (oracle/_ref)")
    if not cffi_available():
        pytest.skip("no interpreter with cffi in this image")
    r = run_cffi_replay(ref_env.REF_LIB, ref_env.ref_assets(), 110)
    assert r.returncode == 0 and "cffi replay ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
