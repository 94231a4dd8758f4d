"""
CPU: the kernel sources themselves (procgen_amd/csrc/pg_env.h + game policies), built with wave.h's host lane-loop
emulation (tests/emu), against the oracle: frames, rew/first/info, entity tables and grids, bit exact, including the
routing between the small and the large LDS arena.
"""
import ctypes as C
import os

import numpy as np
import pytest

import emu_harness
import oracle_env
from helpers import action_stream, assert_rollouts_equal, check_against_generated_assets_fixture, check_against_option_matrix, rollout


def test_rollout_and_state_round_trip_through_the_emulated_kernels():
    """coinrun through the emulated kernels against the oracle, then get_state / set_state into a handle with another seed."""
    n, steps = 8, 200
    acts = action_stream(n, steps, seed=3)
    a = rollout(oracle_env.OracleEnv(n, "coinrun", rand_seed=23), acts)
    emu = emu_harness.EmuEnv(n, "coinrun", rand_seed=23)
    b = rollout(emu, acts)
    assert_rollouts_equal(a, b, "coinrun")
    st = emu.get_state()
    emu2 = emu_harness.EmuEnv(n, "coinrun", rand_seed=99)
    emu2.set_state(st)
    assert emu2.get_state() == st


@pytest.mark.parametrize("game,use_small", [("coinrun", True), ("coinrun", False), ("bigfish", True), ("maze", True), ("climber", True), ("miner", True), ("starpilot", True), ("fruitbot", True), ("leaper", True), ("plunder", True), ("heist", True), ("ninja", True), ("dodgeball", True), ("bossfight", True), ("chaser", True), ("caveflyer", True), ("jumper", True)])
def test_emulated_kernels_match_oracle(game, use_small):
    n, steps = 24, 260
    acts = action_stream(n, steps)
    orc = oracle_env.OracleEnv(n, game, rand_seed=23)
    emu = emu_harness.EmuEnv(n, game, rand_seed=23, use_small=use_small)
    for t in range(steps + 1):
        r1, o1, f1 = orc.observe()
        r2, o2, f2 = emu.observe()
        assert np.array_equal(r1, r2) and np.array_equal(f1, f2), f"step {t}"
        assert np.array_equal(o1["rgb"], o2["rgb"]), f"frame at step {t}"
        for k, v in orc.info_arrays().items():
            assert np.array_equal(v, emu.info_arrays()[k]), f"{k} at step {t}"
        if t % 20 == 0:
            for e in range(n):
                assert np.array_equal(orc.entities(e), emu.entities(e)), f"entities env {e} step {t}"
                assert np.array_equal(orc.grid(e), emu.grid(e)), f"grid env {e} step {t}"
        if t < steps:
            orc.act(acts[t])
            emu.act(acts[t])


def noop_heavy_actions(n, steps, seed, p_noop=0.97):
    """Mostly action 4 (stand still), so episodes live until `cur_time >= timeout` (reference src/game.cpp:134)."""
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(steps):
        a = rng.randint(0, 15, size=(n,), dtype=np.int32)
        a[rng.rand(n) < p_noop] = 4
        out.append(a)
    return out


@pytest.mark.parametrize("game,steps", [("coinrun", 1100)])
def test_long_horizon_over_timeouts_and_generator_twists(game, steps):
    """A rollout long enough that episodes reach their timeout and rand_gen crosses a 624-word block (a twist inside a plain
    step) -- bit-exact against the oracle, entity tables and grids included."""
    n = 8
    acts = noop_heavy_actions(n, steps, seed=21)
    orc = oracle_env.OracleEnv(n, game, rand_seed=23)
    emu = emu_harness.EmuEnv(n, game, rand_seed=23)
    a = rollout(orc, acts)
    b = rollout(emu, acts)
    assert_rollouts_equal(a, b, f"long horizon ({game})")
    for e in range(n):
        assert np.array_equal(orc.entities(e), emu.entities(e)) and np.array_equal(orc.grid(e), emu.grid(e))
    assert a["first"][1000:1002].any(), "an episode must have ended by timeout"


def test_push_recursion_is_pruned_without_changing_a_bit():
    """An object touching several blocking entities is pushed out by each of them at every level of the reference's depth-5
    recursion (BAG:240-268 / 337-369): ~k^5 nested sub_steps that change nothing, 1.3 M wave cycles on the device.  The kernels
    recognise the fixed point and remember no-op calls (pg_env.h push_fixed_point / memo_hit); entity tables must stay identical
    to the oracle's after every step, with the pruning really taken and the nested calls cut several times over."""
    L = emu_harness.lib()
    L.emu_counter.restype = C.c_longlong
    n, steps = 160, 260
    acts = action_stream(n, steps, seed=11)
    orc = oracle_env.OracleEnv(n, "coinrun", rand_seed=23)
    emu = emu_harness.EmuEnv(n, "coinrun", rand_seed=23)
    run0, cut0 = L.emu_counter(5), L.emu_counter(6)
    for t in range(steps):
        orc.act(acts[t])
        emu.act(acts[t])
        r1, o1, f1 = orc.observe()
        r2, o2, f2 = emu.observe()
        assert np.array_equal(r1, r2) and np.array_equal(f1, f2), f"step {t}"
        for e in range(n):
            assert np.array_equal(orc.entities(e), emu.entities(e)), f"entities env {e} step {t}"
    run, cut = L.emu_counter(5) - run0, L.emu_counter(6) - cut0
    assert cut > 1000 and run < 4 * cut, (run, cut)


def test_independent_smart_entities_are_stepped_side_by_side():
    """pg_env.h GameParSmart (coinrun: the agent and the walking enemies): smart entities that nothing can block or reflect
    this step take one lane each in a parallel pass of step_entities.  Heavy levels (many enemies) against the oracle,
    entity tables included, with the pass really taken."""
    L = emu_harness.lib()
    L.emu_counter.restype = C.c_longlong
    before = L.emu_counter(0)
    n, steps = 48, 300
    acts = action_stream(n, steps, seed=19)
    orc = oracle_env.OracleEnv(n, "coinrun", rand_seed=99)
    emu = emu_harness.EmuEnv(n, "coinrun", rand_seed=99)
    for t in range(steps):
        orc.act(acts[t])
        emu.act(acts[t])
        if t % 25 == 0:
            r1, o1, f1 = orc.observe()
            r2, o2, f2 = emu.observe()
            assert np.array_equal(r1, r2) and np.array_equal(f1, f2) and np.array_equal(o1["rgb"], o2["rgb"]), t
            for e in range(n):
                assert np.array_equal(orc.entities(e), emu.entities(e)), (t, e)
    assert L.emu_counter(0) - before > 1000


@pytest.mark.parametrize("game", ["climber", "ninja", "dodgeball", "chaser", "caveflyer"])
def test_parallel_pass_in_the_other_multi_smart_games(game, monkeypatch):
    """The other games that opt into GameParSmart: 300 steps against the oracle with entity tables, the pass taken; and the
    same rollout with the pass switched off (PROCGEN_AMD_DEBUG & 32768) gives the same frames -- the pass commutes."""
    L = emu_harness.lib()
    L.emu_counter.restype = C.c_longlong
    n, steps = 24, 300
    acts = action_stream(n, steps, seed=23)
    before = L.emu_counter(0)
    orc = oracle_env.OracleEnv(n, game, rand_seed=41)
    emu = emu_harness.EmuEnv(n, game, rand_seed=41)
    a, b = rollout(orc, acts), rollout(emu, acts)
    assert_rollouts_equal(a, b, game)
    for e in range(n):
        assert np.array_equal(orc.entities(e), emu.entities(e)), e
    taken = L.emu_counter(0) - before
    assert taken > 500, taken
    monkeypatch.setenv("PROCGEN_AMD_DEBUG", "32768")
    before = L.emu_counter(0)
    assert_rollouts_equal(a, rollout(emu_harness.EmuEnv(n, game, rand_seed=41), acts), game + " (ordered loop only)")
    assert L.emu_counter(0) == before


@pytest.mark.parametrize("game", ["jumper", "caveflyer"])
def test_split_reset_games_hand_ended_episodes_to_the_reset_kernel(game):
    """Games with SPLIT_RESET (pg_env.h GameSplit): their step kernels carry no level generator; an episode that ends is
    finished by the reset kernel (Env::run(2)).  Bit-exact against the oracle, with that path taken."""
    n, steps = 12, 300
    acts = action_stream(n, steps, seed=14)
    emu = emu_harness.EmuEnv(n, game, rand_seed=23)
    assert_rollouts_equal(rollout(oracle_env.OracleEnv(n, game, rand_seed=23), acts), rollout(emu, acts), game)
    assert emu.path_counts()[3] >= 3


@pytest.mark.parametrize("game", ["bossfight", "plunder", "bigfish", "coinrun", "leaper"])
def test_emulated_timeout_classes_from_reference_states(golden_dir, game):
    """The `cur_time >= timeout` path (reference src/game.cpp:134) for every timeout class, from reference states whose
    cur_time was moved next to the timeout (tests/golden/make_timeout_golden.py): kernel logic vs the compiled reference."""
    g = np.load(os.path.join(golden_dir, "timeout_states.npz"))
    n = g[f"{game}/actions"].shape[1]
    emu = emu_harness.EmuEnv(n, game, rand_seed=777)
    emu.set_state([bytes(g[f"{game}/state{e}"]) for e in range(n)])
    got = rollout(emu, list(g[f"{game}/actions"]))
    for k in ("rew", "first", "prev_level_seed", "prev_level_complete", "level_seed", "crc"):
        assert np.array_equal(got[k], g[f"{game}/{k}"]), (game, k)


@pytest.mark.parametrize("game", ["coinrun", "chaser", "dodgeball"])
@pytest.mark.parametrize("kw", [dict(use_monochrome_assets=True), dict(paint_vel_info=True), dict(use_monochrome_assets=True, restrict_themes=True, use_backgrounds=False)])
def test_emulated_option_surface_monochrome_and_vel_info(game, kw):
    """draw_grid_obj / color_for_type fills (use_monochrome_assets) and the velocity squares (paint_vel_info):
    reference src/basic-abstract-game.cpp:455-481,884-886,915-919,960-969."""
    acts = action_stream(6, 80, seed=9)
    a = rollout(oracle_env.OracleEnv(6, game, rand_seed=23, **kw), acts, keep_frames=True)
    b = rollout(emu_harness.EmuEnv(6, game, rand_seed=23, **kw), acts, keep_frames=True)
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_emulated_shard_equals_slice_of_whole():
    """env_offset sharding (include/procgen_amd.h): a shard reproduces the matching slice of one big vector."""
    acts = action_stream(8, 60, seed=3)
    whole = rollout(emu_harness.EmuEnv(8, "coinrun", rand_seed=5), acts)
    shard = rollout(emu_harness.EmuEnv(4, "coinrun", rand_seed=5, env_offset=4), [a[4:] for a in acts])
    for k in whole:
        assert np.array_equal(whole[k][:, 4:], shard[k]), k


def test_emulated_num_levels_and_easy_mode():
    acts = action_stream(4, 80, seed=9)
    for kw in (dict(num_levels=3, start_level=17), dict(distribution_mode=0), dict(use_backgrounds=False), dict(restrict_themes=True), dict(center_agent=False)):
        a = rollout(oracle_env.OracleEnv(4, "coinrun", rand_seed=2, **kw), acts)
        b = rollout(emu_harness.EmuEnv(4, "coinrun", rand_seed=2, **kw), acts)
        assert_rollouts_equal(a, b, str(kw))


@pytest.mark.parametrize("game,mode", [("caveflyer", "memory"), ("maze", "memory"), ("jumper", "memory"), ("dodgeball", "extreme"), ("leaper", "extreme"),
                                       ("heist", "easy"), ("miner", "easy"), ("starpilot", "easy"), ("jumper", "easy")])
def test_emulated_distribution_modes_match_reference_fixture(golden_dir, game, mode):
    """Kernel logic in non-default modes against the compiled reference's fixture (caveflyer memory runs its own 60x60 policy)."""
    g = np.load(os.path.join(golden_dir, "mode_matrix.npz"))
    n = g[f"{game}/{mode}/rew"].shape[1]
    steps = g[f"{game}/{mode}/rew"].shape[0] - 1
    got = rollout(emu_harness.EmuEnv(n, game, rand_seed=23, distribution_mode={"easy": 0, "extreme": 2, "memory": 10}[mode]), action_stream(n, steps))
    for k in ("rew", "first", "level_seed", "crc"):
        assert np.array_equal(got[k], g[f"{game}/{mode}/{k}"]), (game, mode, k)


@pytest.mark.parametrize("game", ["coinrun", "maze", "chaser", "miner"])
def test_emulated_per_cell_grid_path_matches_oracle(monkeypatch, game):
    """The renderer's per-cell blit path for grid cells (taken when a frame cannot use the pull form: adjusted rects,
    more than four image sizes, ...) is forced for every frame (debug flag 1024) and must give the same frames."""
    monkeypatch.setenv("PROCGEN_AMD_DEBUG", "1024")
    n, steps = 6, 60
    acts = action_stream(n, steps, seed=3)
    a = rollout(oracle_env.OracleEnv(n, game, rand_seed=23), acts)
    b = rollout(emu_harness.EmuEnv(n, game, rand_seed=23), acts)
    assert_rollouts_equal(a, b, f"per-cell path ({game})")


@pytest.mark.parametrize("game", ["starpilot", "fruitbot", "coinrun"])
def test_emulated_chunked_entity_path_matches_oracle(monkeypatch, game):
    """Frames with more visible entities than the renderer's register sets hold are drawn chunk by chunk, set up again
    for every band and layer (draw_entities); debug flag 4096 sends every frame down that path (rotated and tiled
    sprites included)."""
    monkeypatch.setenv("PROCGEN_AMD_DEBUG", "4096")
    n, steps = 6, 60
    acts = action_stream(n, steps, seed=5)
    a = rollout(oracle_env.OracleEnv(n, game, rand_seed=23), acts)
    b = rollout(emu_harness.EmuEnv(n, game, rand_seed=23), acts)
    assert_rollouts_equal(a, b, f"chunked entities ({game})")


def test_emulated_option_surface_matches_reference_fixture(golden_dir):
    """A spread of (game, option set) pairs of tests/golden/option_matrix.npz through the emulated kernels (the GPU suite runs all)."""
    g = np.load(os.path.join(golden_dir, "option_matrix.npz"))
    pairs = [("bigfish", "restrict_themes"), ("bossfight", "restrict_themes"), ("fruitbot", "restrict_themes"), ("heist", "restrict_themes"),
             ("plunder", "restrict_themes"), ("maze", "no_backgrounds"), ("starpilot", "no_backgrounds"), ("climber", "no_center_agent"), ("jumper", "no_center_agent"),
             ("ninja", "two_levels"), ("miner", "sequential_levels"), ("dodgeball", "monochrome"), ("caveflyer", "monochrome"), ("leaper", "vel_info")]
    check_against_option_matrix(g, lambda game, n, **kw: emu_harness.EmuEnv(n, game, rand_seed=7, **kw), pairs)


def test_emulated_tall_world_shown_whole_draws_every_background_tile():
    """fruitbot, easy mode, center_agent=False: the 10 x 60 world is scaled into the frame and a dozen background tiles
    (tile_image, reference src/basic-abstract-game.cpp:840-869,999) are on screen at once."""
    n, steps = 4, 50
    acts = action_stream(n, steps, seed=2)
    a = rollout(oracle_env.OracleEnv(n, "fruitbot", rand_seed=3, distribution_mode=0, center_agent=False), acts)
    b = rollout(emu_harness.EmuEnv(n, "fruitbot", rand_seed=3, distribution_mode=0, center_agent=False), acts)
    assert_rollouts_equal(a, b, "fruitbot easy, not centred")


def _actions_with_forced_resets(n, steps, seed):
    rng = np.random.RandomState(seed)
    acts = []
    for _ in range(steps):
        a = rng.randint(0, 15, size=(n,), dtype=np.int32)
        a[rng.rand(n) < 0.05] = -1  # Game::step: action -1 forces a reset (reference src/game.cpp:123-127)
        acts.append(a)
    return acts


@pytest.mark.parametrize("game", ["coinrun", "maze", "starpilot"])
def test_emulated_forced_reset_action(game):
    n, steps = 6, 100
    acts = _actions_with_forced_resets(n, steps, 4)
    a = rollout(oracle_env.OracleEnv(n, game, rand_seed=23), acts)
    b = rollout(emu_harness.EmuEnv(n, game, rand_seed=23), acts)
    assert_rollouts_equal(a, b, f"forced resets ({game})")
    assert a["first"][1:].sum() > 10


def test_product_qt_path_header_matches_qt_pixels(golden_dir):
    """procgen_amd/csrc/pg_qtpath.h -- the code libenv_make builds the jumper compass masks with -- against Qt 5.9.7's own pixels
    (tests/golden/qt_path_ellipses.npz); integer-aligned rects take Qt's midpoint route (pg_render.h exec_ellipse) and are skipped."""
    import ctypes as C

    L = emu_harness.lib()
    L.emu_qt_path_ellipse.argtypes = [C.c_double] * 4 + [C.c_int, C.c_int, C.c_void_p]
    g = np.load(os.path.join(golden_dir, "qt_path_ellipses.npz"))
    unpack = lambda k: np.unpackbits(g[k], axis=-1).astype(bool)
    both_pen, both_brush, brush_only, pen_only = unpack("both"), unpack("both_brush"), unpack("brush_only"), unpack("pen_only")
    out = np.zeros((64, 64), np.uint8)
    checked = 0
    for i, (x, y, w, h) in enumerate(g["rects"]):
        if all(float(int(v)) == v for v in (x, y, w, h)):
            continue
        checked += 1
        L.emu_qt_path_ellipse(x, y, w, h, 1, 1, out.ctypes.data)
        assert np.array_equal(out == 2, both_pen[i]) and np.array_equal(out == 1, both_brush[i]), ("pen + brush", i, (x, y, w, h))
        L.emu_qt_path_ellipse(x, y, w, h, 0, 1, out.ctypes.data)
        assert np.array_equal(out == 1, brush_only[i]), ("brush", i, (x, y, w, h))
        L.emu_qt_path_ellipse(x, y, w, h, 1, 0, out.ctypes.data)
        assert np.array_equal(out == 2, pen_only[i]), ("pen", i, (x, y, w, h))
    assert checked > 1000


@pytest.mark.parametrize("games", [["coinrun", "starpilot"], ["fruitbot", "caveflyer", "chaser"], ["jumper", "bossfight", "leaper", "heist"]])
def test_emulated_generated_assets_match_reference_fixture(golden_dir, games):
    """use_generated_assets=True through the kernel sources: sprites painted on the host at handle creation (pg_assetgen.h), the reset
    path consuming the background generator's draws, the background kernel's painter (pg_bgpaint.h) and the GEN renderer."""
    g = np.load(os.path.join(golden_dir, "generated_assets.npz"))
    check_against_generated_assets_fixture(g, lambda game, n, **kw: emu_harness.EmuEnv(n, game, rand_seed=19, **kw), games)


def test_host_painted_sprites_and_device_style_backgrounds_equal_the_oracles():
    """pg_assetgen.h on the host (what libenv_make uploads) against the oracle's restatement: every object type of four games,
    and a few backgrounds from a generator seeded alike."""
    L = emu_harness.lib()
    O = oracle_env.lib()
    O.pgo_test_generated_asset.argtypes = [C.c_int, C.c_int, C.c_void_p]
    O.pgo_test_generated_background.argtypes = [C.c_int, C.c_void_p]
    L.emu_generated_asset.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
    L.emu_generated_background.argtypes = [C.c_int, C.c_void_p]
    for game in ("coinrun", "leaper", "dodgeball", "miner"):
        gid = O.pgo_game_id(game.encode())
        for t in range(100):
            a = np.zeros(4096, np.uint32)
            b = np.zeros(4096, np.uint32)
            O.pgo_test_generated_asset(gid, t, a.ctypes.data)
            L.emu_generated_asset(game.encode(), t, b.ctypes.data)
            assert np.array_equal(a, b), (game, t)
    for seed in (1, 77, 123456789):
        a = np.zeros(250000, np.uint32)
        b = np.zeros(250000, np.uint32)
        O.pgo_test_generated_background(seed, a.ctypes.data)
        L.emu_generated_background(seed, b.ctypes.data)
        assert np.array_equal(a, b), seed


def test_no_capacity_or_draw_shape_exit_under_the_accepted_option_surface():
    """The renderer's fail(PGE_UNSUPPORTED_DRAW) / capacity exits are not reference semantics and one env reaching one ends the whole
    handle: every (game, distribution_mode, center_agent) the reference accepts, 6 envs x 120 steps, no device error word.  (The
    full sweep -- 64 envs x 2000 steps per configuration -- is tests/tools/draw_limits_sweep.py; DESIGN.md has its last result.)"""
    import importlib.util

    spec = importlib.util.spec_from_file_location("draw_limits_sweep", os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "draw_limits_sweep.py"))
    sweep = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sweep)
    n = 0
    for game, mode, center in sweep.configs():
        assert sweep.run(game, mode, center, 6, 120, 13) == [], (game, mode, center)
        n += 1
    assert n == 2 * (16 * 2 + 4 + 6)


@pytest.mark.parametrize("game,kw", [("coinrun", {}), ("bossfight", {}), ("fruitbot", {}), ("maze", {"distribution_mode": 10}), ("jumper", {"distribution_mode": 0}),
                                     ("caveflyer", {"distribution_mode": 10}), ("chaser", {"distribution_mode": 2}), ("leaper", {}), ("dodgeball", {}), ("starpilot", {})])
def test_kernels_read_no_lds_word_they_did_not_write(monkeypatch, game, kw):
    """LDS is not cleared between workgroups: on a GPU that other processes' kernels share, a workgroup finds whatever the previous one --
    any process's -- left there.  PG_EMU_POISON_LDS fills every emulated workgroup's arena (step, reset and render kernels) with
    pseudo-random words before it runs; a word read before it was written would then show as a mismatch against the oracle.  (Round 4:
    looked for after rare failures that only ever appeared with four test processes sharing the GPU; none found.)"""
    monkeypatch.setenv("PG_EMU_POISON_LDS", "1")
    n, steps = 10, 220
    acts = action_stream(n, steps, seed=13)
    a = rollout(oracle_env.OracleEnv(n, game, rand_seed=29, **kw), acts)
    b = rollout(emu_harness.EmuEnv(n, game, rand_seed=29, **kw), acts)
    assert_rollouts_equal(a, b, f"{game} {kw} on poisoned LDS")


def test_rotation_record_pool_build_variant_draws_the_same_frames():
    """-DPG_ROT_POOL=n (pg_render.h; an experiment for the LDS-bound renderers, default 64 = off): turned / tiled entities that can reach the
    rows being drawn share n records, frames with more of them fall back to the per-band path and cut their chunks into windows.  Built
    with n = 2, so that every fallback runs all the time, the four games with the most turned / tiled sprites must still match the oracle
    bit for bit -- through the default path, the per-band path forced for every frame (debug flag 4096) and without the pull form (1024)."""
    import subprocess
    import sys

    env = dict(os.environ, PG_EMU_GAMES="FruitBot,Dodgeball,StarPilot,Leaper", PG_EMU_DEFS="-DPG_ROT_POOL=2")
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "quick_emu.py")
    for game, extra, dbg in (("fruitbot", [], None), ("dodgeball", [], None), ("starpilot", ["distribution_mode=2"], None), ("leaper", [], "1024"), ("dodgeball", [], "4096"),
                             ("fruitbot", ["center_agent=False"], None)):
        e = dict(env)
        if dbg:
            e["PROCGEN_AMD_DEBUG"] = dbg
        r = subprocess.run([sys.executable, tool, game, "8", "150"] + extra, env=e, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        assert "rotation-record pool:" in r.stdout and " 0 mismatching steps" in r.stdout, r.stdout[-600:]  # (the fallbacks did run)


def test_display_list_kernels_match_the_oracle_through_every_path():
    """The display-list games' frame kernels (pg_prep.h: prep -> raster, the full renderer for queued frames) in the emulation, on poisoned
    LDS, against the oracle: the six games under default options (every frame from its record), the options that send every frame to the full
    renderer's queue, and a build whose register sets hold 8 commands instead of 64 (-DPG_CMD_SET_LANES=8), so that ordinary frames take the
    path a frame with more than 64 visible sprites takes (coinrun: 1 in 10 000, enemy trails) -- commands past the first set fetched per band and
    layer, drawn in the reference's order."""
    import subprocess
    import sys

    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "quick_emu.py")
    base = dict(os.environ, PG_EMU_GAMES="CoinRun,BigFish,Maze,Miner,Climber,Chaser", PG_EMU_POISON_LDS="1")
    base.pop("PG_EMU_DEFS", None)
    cases = [(g, [], None, True) for g in ("coinrun", "bigfish", "maze", "miner", "climber", "chaser")]
    cases += [("coinrun", ["center_agent=False"], None, False), ("coinrun", ["use_monochrome_assets=True"], None, False), ("coinrun", ["paint_vel_info=True"], None, False),
              ("climber", ["use_backgrounds=False"], None, True)]
    cases += [(g, [], "-DPG_CMD_SET_LANES=8", True) for g in ("coinrun", "bigfish", "chaser")]
    for game, extra, defs, fast in cases:
        e = dict(base)
        if defs:
            e["PG_EMU_DEFS"] = defs
        r = subprocess.run([sys.executable, tool, game, "8", "120"] + extra, env=e, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and " 0 mismatching steps" in r.stdout, (game, extra, defs, r.stdout[-1200:] + r.stderr[-1200:])
        import re
        m = re.search(r"frames drawn from their record (\d+) / by the full renderer (\d+)", r.stdout)
        assert m, r.stdout[-600:]
        rec, full = int(m.group(1)), int(m.group(2))
        assert (rec > 0 and full <= rec) if fast else (rec == 0 and full > 0), (game, extra, rec, full)
