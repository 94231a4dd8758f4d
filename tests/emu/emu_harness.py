"""
TEST HARNESS: Python wrapper around tests/emu/libemu.so (the kernel sources built with wave.h's host lane-loop
emulation).  Mirrors the observe()/act() shape of the libenv mirror so the same comparison helpers apply.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
# PG_EMU_GAMES="CoinRun,BigFish": a quick build of a few policies while iterating on the kernels (its own file; the tests use the full one)
_SUBSET = os.environ.get("PG_EMU_GAMES", "")
# PG_EMU_DEFS="-DPG_ROT_POOL=2": the kernel sources with build-time experiment switches (its own file again)
_DEFS = os.environ.get("PG_EMU_DEFS", "").split()
_TAG = ("_" + _SUBSET.replace(",", "_") if _SUBSET else "") + "".join("_" + "".join(ch for ch in d if ch.isalnum()) for d in _DEFS)
LIB = os.path.join(HERE, "libemu" + _TAG + ".so")
CSRC = os.path.join(REPO, "procgen_amd", "csrc")


def build(force=False):
    srcs = [os.path.join(HERE, "emu_env.cpp"), os.path.join(CSRC, "assets.cpp"), os.path.join(CSRC, "image_io.cpp"), os.path.join(CSRC, "state_io.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    def fresh():
        return os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps)

    if not force and fresh():
        return
    import fcntl

    with open(LIB + ".lock", "w") as lock:  # pytest-xdist workers: one builds, the others wait and then find it fresh
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or not fresh():
            tmp = LIB + f".tmp{os.getpid()}"
            subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-march=ivybridge", "-fno-strict-aliasing", "-fPIC",
                                   "-shared", "-I" + CSRC] + _DEFS + (["-DPG_HUMAN_TRACE"] if os.environ.get("PG_HUMAN_TRACE") else [])
                                  + (["-DPG_FOR_EACH_GAME(X)=" + " ".join(f"X({g})" for g in _SUBSET.split(","))] if _SUBSET else []) + srcs + ["-lz", "-o", tmp])
            os.replace(tmp, LIB)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        L.emu_make.restype = C.c_void_p
        L.emu_make.argtypes = [C.c_char_p] + [C.c_int] * 11 + [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.emu_free.argtypes = [C.c_void_p]
        L.emu_init.argtypes = [C.c_void_p]
        L.emu_step.argtypes = [C.c_void_p, C.c_void_p]
        L.emu_observe.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        for f in ("emu_error", "emu_num_entities", "emu_is_big"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_int]
        L.emu_get_state.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]
        L.emu_set_state.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]
        L.emu_render_human.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.emu_path_counts.argtypes = [C.c_void_p, C.c_void_p]
        L.emu_dump_entities.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.emu_dump_grid.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _lib = L
    return _lib


class EmuEnv:
    def __init__(self, num, env_name, rand_seed=0, env_offset=0, num_levels=0, start_level=0, distribution_mode=1,
                 center_agent=True, use_backgrounds=True, restrict_themes=False, use_sequential_levels=False, debug_mode=0,
                 resource_root=None, atlas_path=None, use_small=True, use_monochrome_assets=False, paint_vel_info=False, use_generated_assets=False):
        self.L = lib()
        self.num = num
        if atlas_path is None:
            atlas_path = os.path.join(REPO, "procgen_amd", "data", f"{env_name}.atlas")
        if resource_root is None:
            resource_root = "/root/reference/procgen/data/assets/"
        self.h = C.c_void_p(self.L.emu_make(env_name.encode(), num, rand_seed, env_offset, num_levels, start_level, distribution_mode,
                                            int(center_agent), int(use_backgrounds), int(restrict_themes), int(use_sequential_levels),
                                            debug_mode, resource_root.encode(), atlas_path.encode(), int(use_small), int(use_monochrome_assets), int(paint_vel_info), int(use_generated_assets)))
        assert self.h, "emu_make failed"
        self.rgb = np.zeros((num, 64, 64, 3), np.uint8)
        self.rew = np.zeros(num, np.float32)
        self.first = np.zeros(num, np.uint8)
        self.info = {"prev_level_seed": np.zeros(num, np.int32), "prev_level_complete": np.zeros(num, np.uint8),
                     "level_seed": np.zeros(num, np.int32)}
        self.L.emu_init(self.h)

    def observe(self):
        i = self.info
        self.L.emu_observe(self.h, self.rgb.ctypes.data, self.rew.ctypes.data, self.first.ctypes.data,
                           i["prev_level_seed"].ctypes.data, i["prev_level_complete"].ctypes.data, i["level_seed"].ctypes.data)
        for e in range(self.num):
            err = self.L.emu_error(self.h, e)
            assert err == 0, f"kernel error flag {err} in env {e}"
        return self.rew, {"rgb": self.rgb}, self.first.astype(bool)

    def act(self, ac):
        ac = np.ascontiguousarray(ac, dtype=np.int32)
        self.L.emu_step(self.h, ac.ctypes.data)

    def info_arrays(self):
        return self.info

    def render_human(self, env):
        """the 512 x 512 x 3 render_human frame of one env (pg_human.h), drawn from the env's current state"""
        out = np.zeros((512, 512, 3), np.uint8)
        assert self.L.emu_render_human(self.h, env, out.ctypes.data) == 0
        return out

    def entities(self, env):
        n = self.L.emu_num_entities(self.h, env)
        out = np.zeros((n, 31), np.int32)
        if n:
            self.L.emu_dump_entities(self.h, env, out.ctypes.data)
        return out

    def grid(self, env):
        out = np.zeros(64 * 64, np.int32)
        w, h = C.c_int(), C.c_int()
        self.L.emu_dump_grid(self.h, env, out.ctypes.data, C.byref(w), C.byref(h))
        return out[: w.value * h.value].reshape(h.value, w.value)

    def get_state(self):
        buf = C.create_string_buffer(1 << 20)
        out = []
        for e in range(self.num):
            n = self.L.emu_get_state(self.h, e, buf, 1 << 20)
            assert n > 0
            out.append(bytes(buf.raw[:n]))
        return out

    def set_state(self, states):
        for e, st in enumerate(states):
            assert self.L.emu_set_state(self.h, e, st, len(st)) == 0

    def path_counts(self):
        """(0, 0, env-steps taken by the step
        kernels, episodes a SPLIT_RESET game's step kernels handed to the reset kernel)"""
        out = (C.c_longlong * 4)()
        self.L.emu_path_counts(self.h, out)
        return tuple(out)

    def is_big(self, env):
        return self.L.emu_is_big(self.h, env)

    def close(self):
        if self.h:
            self.L.emu_free(self.h)
            self.h = None
