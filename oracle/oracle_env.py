"""
TEST INFRASTRUCTURE: Python harness around the plain-C oracle (oracle/libprocgen_oracle.so).

Image pixels are handed to the oracle from outside (it never decodes files): either decoded here from the
PNG tree with Pillow + Qt's premultiply formula (independent of the product's decoder), or read from the
product's baked .atlas pack when the PNG tree is absent (the GPU box).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libprocgen_oracle.so")
REPO = os.path.dirname(HERE)


class PgoOptions(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "rand_seed", "num_levels", "start_level", "distribution_mode", "center_agent", "use_backgrounds",
        "use_monochrome_assets", "restrict_themes", "use_generated_assets", "paint_vel_info",
        "use_sequential_levels", "debug_mode")]


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "port"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        L.pgo_game_id.argtypes = [C.c_char_p]
        L.pgo_image_name.restype = C.c_char_p
        L.pgo_image_name.argtypes = [C.c_int, C.c_int]
        L.pgo_set_image.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.pgo_make.restype = C.c_void_p
        L.pgo_make.argtypes = [C.c_int, C.c_int, C.POINTER(PgoOptions)]
        L.pgo_make_strided.restype = C.c_void_p
        L.pgo_make_strided.argtypes = [C.c_int, C.c_int, C.POINTER(PgoOptions), C.c_int, C.c_int]
        L.pgo_free.argtypes = [C.c_void_p]
        L.pgo_init.argtypes = [C.c_void_p]
        L.pgo_step.argtypes = [C.c_void_p, C.c_void_p]
        L.pgo_observe.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.pgo_num_entities.argtypes = [C.c_void_p, C.c_int]
        L.pgo_dump_entities.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.pgo_dump_grid.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.pgo_dump_scalars.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def qt_premultiply(argb):
    """Qt's qPremultiply() (qrgb.h) on an array of 0xAARRGGBB words."""
    x = argb.astype(np.uint64)
    a = x >> 24
    t = (x & 0xff00ff) * a
    t = ((t + ((t >> 8) & 0xff00ff) + 0x800080) >> 8) & 0xff00ff
    g = ((x >> 8) & 0xff) * a
    g = (g + ((g >> 8) & 0xff) + 0x80) & 0xff00
    return (g | t | (a << 24)).astype(np.uint32)


def decode_png_qt(path, is_bg):
    """QImage(path).convertToFormat(ARGB32_Premultiplied | RGB32) with Pillow as the PNG decoder."""
    from PIL import Image

    im = Image.open(path)
    rgba = np.asarray(im.convert("RGBA"), dtype=np.uint32)
    argb = (rgba[..., 3] << 24) | (rgba[..., 0] << 16) | (rgba[..., 1] << 8) | rgba[..., 2]
    if is_bg:
        return (argb | 0xff000000).astype(np.uint32)
    return qt_premultiply(argb)


_images_loaded = set()


def default_atlas(game):
    return os.path.join(REPO, "procgen_amd", "data", f"{game}.atlas")


def load_images(game, resource_root=None, atlas_path=None):
    L = lib()
    gid = L.pgo_game_id(game.encode())
    assert gid >= 0, f"game {game} not restated in the oracle"
    if gid in _images_loaded:
        return gid
    if resource_root is None and os.path.isdir("/root/reference/procgen/data/assets"):
        resource_root = "/root/reference/procgen/data/assets"
    pack = None
    if resource_root is None:
        from procgen_amd.atlas import read_atlas

        pack = read_atlas(atlas_path or default_atlas(game))
    for i in range(L.pgo_num_images(gid)):
        name = L.pgo_image_name(gid, i).decode()
        is_bg = L.pgo_image_is_background(gid, i)
        if pack is not None:
            fmt, px = pack[name + ("|bg" if is_bg else "")]
        else:
            px = decode_png_qt(os.path.join(resource_root, name), is_bg)
        px = np.ascontiguousarray(px, dtype=np.uint32)
        L.pgo_set_image(gid, i, px.shape[1], px.shape[0], px.ctypes.data)
    _images_loaded.add(gid)
    return gid


class OracleEnv:
    """Vectorized oracle env with the observe()/act() shape of the libenv mirror."""

    def __init__(self, num, env_name, rand_seed=0, num_levels=0, start_level=0, distribution_mode=1,
                 center_agent=True, use_backgrounds=True, use_monochrome_assets=False, restrict_themes=False,
                 paint_vel_info=False, use_sequential_levels=False, debug_mode=0, resource_root=None, atlas_path=None,
                 env_offset=0, env_stride=1, use_generated_assets=False):
        self.L = lib()
        self.gid = load_images(env_name, resource_root, atlas_path)
        self.num = num
        o = PgoOptions(rand_seed, num_levels, start_level, distribution_mode, int(center_agent), int(use_backgrounds),
                       int(use_monochrome_assets), int(restrict_themes), int(use_generated_assets), int(paint_vel_info),
                       int(use_sequential_levels), debug_mode)
        self.h = C.c_void_p(self.L.pgo_make_strided(self.gid, num, C.byref(o), env_offset, env_stride))
        self.rgb = np.zeros((num, 64, 64, 3), np.uint8)
        self.rew = np.zeros(num, np.float32)
        self.first = np.zeros(num, np.uint8)
        self.info = {"prev_level_seed": np.zeros(num, np.int32), "prev_level_complete": np.zeros(num, np.uint8),
                     "level_seed": np.zeros(num, np.int32)}
        self.L.pgo_init(self.h)

    def observe(self):
        i = self.info
        self.L.pgo_observe(self.h, self.rgb.ctypes.data, self.rew.ctypes.data, self.first.ctypes.data,
                           i["prev_level_seed"].ctypes.data, i["prev_level_complete"].ctypes.data, i["level_seed"].ctypes.data)
        return self.rew, {"rgb": self.rgb}, self.first.astype(bool)

    def act(self, ac):
        ac = np.ascontiguousarray(ac, dtype=np.int32)
        assert ac.shape == (self.num,)
        self.L.pgo_step(self.h, ac.ctypes.data)

    def info_arrays(self):
        return self.info

    def entities(self, env):
        n = self.L.pgo_num_entities(self.h, env)
        out = np.zeros((n, 31), np.int32)
        if n:
            self.L.pgo_dump_entities(self.h, env, out.ctypes.data)
        return out

    def grid(self, env):
        out = np.zeros(64 * 64, np.int32)
        w, h = C.c_int(), C.c_int()
        self.L.pgo_dump_grid(self.h, env, out.ctypes.data, C.byref(w), C.byref(h))
        return out[: w.value * h.value].reshape(h.value, w.value)

    def scalars(self, env):
        out = np.zeros(16, np.int32)
        self.L.pgo_dump_scalars(self.h, env, out.ctypes.data)
        return out

    def close(self):
        if self.h:
            self.L.pgo_free(self.h)
            self.h = None
