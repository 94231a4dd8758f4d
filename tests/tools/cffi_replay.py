"""Binds libenv.so through **cffi** -- the FFI gym3's CEnv uses (reference procgen/env.py:66,128-136) -- and replays a golden rollout.

Runs under an interpreter that has cffi (this image: /opt/conda/bin/python3.9; the system python has none), as a subprocess of
tests/test_gpu_parity.py::test_boundary_through_cffi.  The call sequence is gym3.libenv.CEnv's: cdef of libenv.h plus the reference's two
`c_func_defs` strings, ffi.dlopen, libenv_make(num, options) with `struct libenv_options` BY VALUE (reference src/vecgame.cpp:47-50),
libenv_get_tensortypes x 3, numpy buffers, libenv_set_buffers, observe / act, get_state through call_c_func's path, libenv_close.

usage: cffi_replay.py <libenv.so> <include dir> <golden .npz> <resource_root> [steps]
prints "cffi replay ok: ..." and exits 0, or the first mismatch and exits 1.
"""
import re
import sys
import zlib

import numpy as np
from cffi import FFI

lib_path, include_dir, golden_path, resource_root = sys.argv[1:5]
max_steps = int(sys.argv[5]) if len(sys.argv) > 5 else 10**9

# the header as gym3 feeds it to cffi: preprocessor lines, the extern "C" braces and the export macro removed
src = open(f"{include_dir}/libenv.h").read()
src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
lines = []
for line in src.splitlines():
    s = line.strip()
    if s.startswith("#") or s in ('extern "C" {', "}"):
        continue
    lines.append(line.replace("LIBENV_API", ""))
cdef = "\n".join(lines)
cdef = cdef.replace("LIBENV_MAX_NAME_LEN", "128").replace("LIBENV_MAX_NDIM", "16")
# the reference's own c_func_defs, verbatim (procgen/env.py:132-135)
C_FUNC_DEFS = [
    "int get_state(libenv_env *, int, char *, int);",
    "void set_state(libenv_env *, int, char *, int);",
]
ffi = FFI()
ffi.cdef(cdef + "\n" + "\n".join(C_FUNC_DEFS))
c = ffi.dlopen(lib_path)
assert c.libenv_version() == 1

gold = np.load(golden_path)
acts = gold["actions"]
num = acts.shape[1]
steps = min(acts.shape[0] - 1, max_steps)

# options, encoded as gym3 does: bool -> uint8[1], int -> int32[1], str -> uint8[len] (not NUL-terminated)
options = {
    "center_agent": True, "use_generated_assets": False, "use_monochrome_assets": False, "restrict_themes": False,
    "use_backgrounds": True, "paint_vel_info": False, "distribution_mode": 1,  # "hard"
    "env_name": "coinrun", "num_levels": 0, "start_level": 0, "num_actions": 15, "use_sequential_levels": False,
    "debug_mode": 0, "rand_seed": 23, "num_threads": 4, "render_human": False, "resource_root": resource_root,
}
keep = []
c_items = ffi.new("struct libenv_option[]", len(options))
for i, (k, v) in enumerate(options.items()):
    name = k.encode()
    c_items[i].name = name
    if isinstance(v, bool):
        arr = np.array([v], dtype=np.uint8)
        c_items[i].dtype = c.LIBENV_DTYPE_UINT8
    elif isinstance(v, int):
        arr = np.array([v], dtype=np.int32)
        c_items[i].dtype = c.LIBENV_DTYPE_INT32
    else:
        arr = np.frombuffer(v.encode(), dtype=np.uint8).copy()
        c_items[i].dtype = c.LIBENV_DTYPE_UINT8
    keep.append(arr)
    c_items[i].count = arr.size
    c_items[i].data = ffi.cast("void *", arr.ctypes.data)
c_options = ffi.new("struct libenv_options *")
c_options.items = c_items
c_options.count = len(options)
handle = c.libenv_make(num, c_options[0])  # by value
assert handle != ffi.NULL

NP = {c.LIBENV_DTYPE_UINT8: np.uint8, c.LIBENV_DTYPE_INT32: np.int32, c.LIBENV_DTYPE_FLOAT32: np.float32}


def tensortypes(space):
    count = c.libenv_get_tensortypes(handle, space, ffi.NULL)
    tt = ffi.new("struct libenv_tensortype[]", max(count, 1))
    c.libenv_get_tensortypes(handle, space, tt)
    return [(ffi.string(tt[i].name).decode(), NP[tt[i].dtype], tuple(tt[i].shape[j] for j in range(tt[i].ndim))) for i in range(count)]


ob_t, ac_t, info_t = tensortypes(c.LIBENV_SPACE_OBSERVATION), tensortypes(c.LIBENV_SPACE_ACTION), tensortypes(c.LIBENV_SPACE_INFO)
assert [t[0] for t in ob_t] == ["rgb"] and ob_t[0][2] == (64, 64, 3), ob_t
assert [t[0] for t in ac_t] == ["action"], ac_t
assert sorted(t[0] for t in info_t) == ["level_seed", "prev_level_complete", "prev_level_seed"], info_t


def alloc(types):
    arrays = {name: np.zeros((num,) + shape, dtype=dt) for name, dt, shape in types}
    tab = ffi.new("void *[]", max(len(types) * num, 1))
    for s, (name, _, _) in enumerate(types):
        a = arrays[name]
        for e in range(num):
            tab[s * num + e] = ffi.cast("void *", a.ctypes.data + e * a.strides[0])
    return arrays, tab


ob, ob_tab = alloc(ob_t)
ac, ac_tab = alloc(ac_t)
info, info_tab = alloc(info_t)
rew = np.zeros(num, dtype=np.float32)
first = np.zeros(num, dtype=np.uint8)
bufs = ffi.new("struct libenv_buffers *")
bufs.ob = ob_tab
bufs.ac = ac_tab
bufs.info = info_tab
bufs.rew = ffi.cast("float *", rew.ctypes.data)
bufs.first = ffi.cast("uint8_t *", first.ctypes.data)
c.libenv_set_buffers(handle, bufs)

state_checks = 0
for t in range(steps + 1):
    c.libenv_observe(handle)
    crc = np.array([zlib.crc32(ob["rgb"][e].tobytes()) for e in range(num)], dtype=np.uint32)
    for k, got in (("rew", rew), ("first", first), ("level_seed", info["level_seed"]), ("prev_level_seed", info["prev_level_seed"]),
                   ("prev_level_complete", info["prev_level_complete"]), ("crc", crc)):
        if not np.array_equal(np.asarray(got).astype(gold[k].dtype), gold[k][t]):
            print(f"cffi replay: {k} differs at step {t}")
            sys.exit(1)
    for e in (0, 1):  # the reference's get_state through cffi: char[] buffer, bytes of the golden state
        key = f"state{t}_e{e}_bytes"
        if key in gold.files:
            length = 2**20
            buf = ffi.new(f"char[{length}]")
            n = c.get_state(handle, e, buf, length)
            if bytes(ffi.buffer(buf, n)) != gold[key].tobytes():
                print(f"cffi replay: get_state of env {e} differs at step {t}")
                sys.exit(1)
            state_checks += 1
    if t < steps:
        ac["action"][...] = acts[t]
        c.libenv_act(handle)
# set_state through cffi (bytes -> char *), then the frame is the restored one
if "state0_e0_bytes" in gold.files:
    st = gold["state0_e0_bytes"].tobytes()
    c.set_state(handle, 0, st, len(st))
    c.libenv_observe(handle)
    if zlib.crc32(ob["rgb"][0].tobytes()) != int(gold["crc"][0][0]):
        print("cffi replay: frame after set_state differs")
        sys.exit(1)
c.libenv_close(handle)
print(f"cffi replay ok: {num} envs x {steps} steps, {state_checks} states, cffi {__import__('cffi').__version__}")
