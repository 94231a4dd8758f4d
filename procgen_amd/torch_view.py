"""
Zero-copy torch views of the library's device buffers (the hook is the reference-sanctioned
``c_func_defs`` / ``call_c_func`` route of reference procgen/env.py:132-135,145: extra exported functions on the libenv
handle -- here procgen_amd_part_buffers, include/procgen_amd.h).

    env = ProcgenGym3Env(num=65536, env_name="coinrun", extra_options={"host_observations": False})
    views = device_views(env)              # one DeviceView per part (a single-game, single-device handle has one)
    env.act(actions); env.observe()        # libenv_observe joins the library's stream: the views now hold this step
    obs = views[0].ob                      # torch.uint8 [n, 64, 64, 3] on cuda:<device>, ALIASING the library's buffer

Nothing is copied: the tensors wrap the HBM arrays the render / step kernels write (``__cuda_array_interface__`` v3, which
torch's ROCm build consumes as on CUDA), so a policy reads observations where they were rasterized.  They stay valid until
``env.close()``.  The contents are those of the last step once ``env.observe()`` has returned (it synchronizes the producing
stream); a consumer that wants to overlap may instead make its own stream wait on ``DeviceView.stream`` -- the raw
hipStream_t the kernels run on.

A handle is G device shards x K games parts (include/procgen_amd.h): part p holds its envs densely, its env i is global env
``first_env + i * env_stride``; ``global_indices()`` gives them, ``scatter_observations`` assembles one [N, 64, 64, 3]
tensor on a chosen device for callers that want the gym3 layout (that one copies, by definition).

PyTorch is plumbing here: this module is the only file of the package that imports it, and only when called.
"""
import ctypes as C

import numpy as np


class _Buffers(C.Structure):  # struct procgen_amd_buffers
    _fields_ = [("device_id", C.c_int), ("num_envs", C.c_int), ("stream", C.c_void_p), ("ob", C.c_void_p), ("rew", C.c_void_p),
                ("first", C.c_void_p), ("prev_level_seed", C.c_void_p), ("prev_level_complete", C.c_void_p), ("level_seed", C.c_void_p),
                ("action", C.c_void_p)]


class _Part(C.Structure):  # struct procgen_amd_part
    _fields_ = [("buffers", _Buffers), ("first_env", C.c_int), ("env_stride", C.c_int), ("game", C.c_char * 128)]


class _DevArray:
    """the minimum torch.as_tensor needs to adopt device memory it does not own"""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 3, "strides": None}
        self._owner = owner  # keeps the env (and so the allocation) alive as long as a tensor built from this object is


def _tensor(ptr, shape, typestr, device_id, owner):
    import torch

    return torch.as_tensor(_DevArray(ptr, shape, typestr, owner), device=torch.device("cuda", device_id))


class DeviceView:
    """torch tensors over one part's device arrays (no copies)."""

    def __init__(self, env, part):
        b = part.buffers
        n = b.num_envs
        self.device_id = b.device_id
        self.num_envs = n
        self.stream = b.stream  # raw hipStream_t of the producing kernels
        self.first_env = part.first_env
        self.env_stride = part.env_stride
        self.game = part.game.decode()
        self.ob = _tensor(b.ob, (n, 64, 64, 3), "|u1", b.device_id, env)
        self.rew = _tensor(b.rew, (n,), "<f4", b.device_id, env)
        self.first = _tensor(b.first, (n,), "|u1", b.device_id, env)
        self.prev_level_seed = _tensor(b.prev_level_seed, (n,), "<i4", b.device_id, env)
        self.prev_level_complete = _tensor(b.prev_level_complete, (n,), "|u1", b.device_id, env)
        self.level_seed = _tensor(b.level_seed, (n,), "<i4", b.device_id, env)
        self.action = _tensor(b.action, (n,), "<i4", b.device_id, env)

    def global_indices(self):
        return self.first_env + self.env_stride * np.arange(self.num_envs)


def device_views(env):
    """[DeviceView] -- one per part of the handle, in part order."""
    lib, handle = env._lib, env._handle
    lib.procgen_amd_part_buffers.argtypes = [C.c_void_p, C.POINTER(_Part), C.c_int]
    lib.procgen_amd_part_buffers.restype = C.c_int
    count = lib.procgen_amd_part_buffers(handle, None, 0)
    parts = (_Part * count)()
    assert lib.procgen_amd_part_buffers(handle, parts, count) == count
    return [DeviceView(env, p) for p in parts]


def device_observations(env):
    """The observation tensor of a single-part handle: torch.uint8 [N, 64, 64, 3] aliasing the library's buffer."""
    views = device_views(env)
    if len(views) != 1:
        raise ValueError(f"this handle has {len(views)} parts (games x devices): use device_views(env), or scatter_observations(env)")
    return views[0].ob


def scatter_observations(env, device=None, out=None):
    """One [N, 64, 64, 3] uint8 tensor in global env order from all parts (copies: parts are strided in the global order and may
    sit on different devices).  device: where to assemble (default: the first part's)."""
    import torch

    views = device_views(env)
    dev = torch.device(device) if device is not None else torch.device("cuda", views[0].device_id)
    if out is None:
        out = torch.empty((env.num, 64, 64, 3), dtype=torch.uint8, device=dev)
    for v in views:
        idx = torch.as_tensor(v.global_indices(), device=dev)
        out.index_copy_(0, idx, v.ob.to(dev, non_blocking=True))
    return out
