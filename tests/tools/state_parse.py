"""
Parser for the reference's get_state wire format (reference src/vecgame.cpp:437-445, src/game.cpp:170-229,
src/basic-abstract-game.cpp:1152-1223, src/entity.cpp:90-137, src/randgen.cpp:100-107, src/buffer.h).
Little-endian int32 / float32 stream; strings are (int32 length, bytes); RNGs are the iostream text of mt19937.
"""
import struct

import numpy as np

ENTITY_FIELDS = [
    ("x", "f"), ("y", "f"), ("vx", "f"), ("vy", "f"), ("rx", "f"), ("ry", "f"),
    ("type", "i"), ("image_type", "i"), ("image_theme", "i"), ("render_z", "i"),
    ("will_erase", "i"), ("collides_with_entities", "i"),
    ("collision_margin", "f"), ("rotation", "f"), ("vrot", "f"),
    ("is_reflected", "i"), ("fire_time", "i"), ("spawn_time", "i"), ("life_time", "i"), ("expire_time", "i"),
    ("use_abs_coords", "i"),
    ("friction", "f"), ("smart_step", "i"), ("avoids_collisions", "i"), ("auto_erase", "i"),
    ("alpha", "f"), ("health", "f"), ("theta", "f"), ("grow_rate", "f"), ("alpha_decay", "f"), ("climber_spawn_x", "f"),
]


class Reader:
    def __init__(self, data):
        self.d = data
        self.o = 0

    def i(self):
        (v,) = struct.unpack_from("<i", self.d, self.o)
        self.o += 4
        return v

    def f(self):
        (v,) = struct.unpack_from("<f", self.d, self.o)
        self.o += 4
        return np.float32(v)

    def s(self):
        n = self.i()
        v = self.d[self.o:self.o + n]
        self.o += n
        return v

    def rng(self):
        seeded = self.i()
        txt = self.s().decode().split()
        return {"is_seeded": seeded, "mt": np.array(txt[:624], dtype=np.uint64).astype(np.uint32), "idx": int(txt[624])}


def parse_state(data, game_extra=None):
    r = Reader(data)
    st = {}
    st["version"] = r.i()
    st["game_name"] = r.s().decode()
    for k in ("paint_vel_info", "use_generated_assets", "use_monochrome_assets", "restrict_themes", "use_backgrounds",
              "center_agent", "debug_mode", "distribution_mode", "use_sequential_levels", "use_easy_jump", "plain_assets",
              "physics_mode", "grid_step", "level_seed_low", "level_seed_high", "game_type", "game_n"):
        st[k] = r.i()
    st["level_seed_rand_gen"] = r.rng()
    st["rand_gen"] = r.rng()
    st["reward"] = r.f()
    for k in ("done", "level_complete", "action", "timeout", "current_level_seed", "prev_level_seed",
              "episodes_remaining", "episode_done", "last_reward_timer"):
        st[k] = r.i()
    st["last_reward"] = r.f()
    for k in ("default_action", "fixed_asset_seed", "cur_time", "is_waiting_for_step"):
        st["offset_of_" + k] = r.o  # byte offset of the field (tests/golden/make_timeout_golden.py patches cur_time)
        st[k] = r.i()
    # BasicAbstractGame
    st["grid_size"] = r.i()
    n = r.i()
    ents = []
    for _ in range(n):
        ents.append({name: (r.f() if t == "f" else r.i()) for name, t in ENTITY_FIELDS})
    st["entities"] = ents
    st["use_procgen_background"] = r.i()
    st["background_index"] = r.i()
    st["bg_tile_ratio"] = r.f()
    st["bg_pct_x"] = r.f()
    st["char_dim"] = r.f()
    for k in ("last_move_action", "move_action", "special_action"):
        st[k] = r.i()
    for k in ("mixrate", "maxspeed", "max_jump", "action_vx", "action_vy", "action_vrot", "center_x", "center_y"):
        st[k] = r.f()
    for k in ("random_agent_start", "has_useful_vel_info", "step_rand_int"):
        st[k] = r.i()
    st["asset_rand_gen"] = r.rng()
    for k in ("main_width", "main_height", "out_of_bounds_object"):
        st[k] = r.i()
    for k in ("unit", "view_dim", "x_off", "y_off", "visibility", "min_visibility"):
        st[k] = r.f()
    gw, gh = r.i(), r.i()
    cnt = r.i()
    st["grid"] = np.frombuffer(data, dtype="<i4", count=cnt, offset=r.o).reshape(gh, gw).copy()
    r.o += 4 * cnt
    st["extra_offset"] = r.o
    if st["game_name"] == "coinrun":  # reference src/games/coinrun.cpp:500-509
        st["last_agent_y"] = r.f()
        st["wall_theme"] = r.i()
        st["has_support"] = r.i()
        st["facing_right"] = r.i()
        st["is_on_crate"] = r.i()
        st["gravity"] = r.f()
        st["air_control"] = r.f()
        end = r.i()
        assert (end & 0xFFFFFFFF) == 0xCAFECAFE, hex(end)
    return st


def entities_as_words(st):
    """(n, 31) int32 array with floats bit-cast, same layout as the oracle's pgo_dump_entities."""
    out = np.zeros((len(st["entities"]), 31), np.int32)
    for i, e in enumerate(st["entities"]):
        for k, (name, t) in enumerate(ENTITY_FIELDS):
            out[i, k] = np.float32(e[name]).view(np.int32) if t == "f" else e[name]
    return out
