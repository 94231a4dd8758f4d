"""
Host-side mirror of the reference's Python environment classes
(reference procgen/env.py:66-246: BaseProcgenEnv / ProcgenGym3Env).

Same constructor arguments, same option dictionary, same action combos and the same
get_state / set_state / act / observe surface -- but ``lib_dir`` defaults to this
package's HIP ``libenv.so`` (procgen_amd/csrc/build/libenv.so) and the cffi ``CEnv`` of gym3 is
replaced by the ctypes binding in procgen_amd/libenv.py.  Nothing here computes game
state: every call goes through the libenv C ABI (include/libenv.h).
"""
import os
import random

import numpy as np

from .libenv import CEnv

SCRIPT_DIR = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB_DIR = os.environ.get("PROCGEN_AMD_LIB_DIR") or os.path.join(SCRIPT_DIR, "csrc", "build")

MAX_STATE_SIZE = 2 ** 20

# reference procgen/env.py:14-31
ENV_NAMES = [
    "bigfish", "bossfight", "caveflyer", "chaser", "climber", "coinrun", "dodgeball", "fruitbot",
    "heist", "jumper", "leaper", "maze", "miner", "ninja", "plunder", "starpilot",
]

# reference procgen/env.py:33-42
EXPLORATION_LEVEL_SEEDS = {
    "coinrun": 1949448038, "caveflyer": 1259048185, "leaper": 1318677581, "jumper": 1434825276,
    "maze": 158988835, "heist": 876640971, "climber": 1561126160, "ninja": 1123500215,
}

# reference procgen/env.py:45-51
DISTRIBUTION_MODE_DICT = {"easy": 0, "hard": 1, "extreme": 2, "memory": 10, "exploration": 20}


def create_random_seed():
    """reference procgen/env.py:54-63 (mpi4py rank mixing kept: it only touches the seed)."""
    rand_seed = random.SystemRandom().randint(0, 2 ** 31 - 1)
    try:
        from mpi4py import MPI

        rand_seed = rand_seed - (rand_seed % MPI.COMM_WORLD.size) + MPI.COMM_WORLD.rank
    except ModuleNotFoundError:
        pass
    return rand_seed


def default_resource_root():
    """Directory of PNG assets.  The HIP library prefers its baked atlas (procgen_amd/data/*.atlas)
    and only decodes PNGs from here when no atlas is present; the path is still passed, as the
    reference does (reference procgen/env.py:86-88,122)."""
    for cand in (os.environ.get("PROCGEN_RESOURCE_ROOT"),
                 os.path.join(SCRIPT_DIR, "data", "assets"),
                 "/root/reference/procgen/data/assets"):
        if cand and os.path.isdir(cand):
            return cand.rstrip(os.sep) + os.sep
    return ""


class BaseProcgenEnv(CEnv):
    def __init__(self, num, env_name, options, debug=False, rand_seed=None, num_levels=0, start_level=0,
                 use_sequential_levels=False, debug_mode=0, resource_root=None, num_threads=4,
                 render_mode=None, lib_dir=None, extra_options=None, buffer_padding=0):
        if resource_root is None:
            resource_root = default_resource_root()
        if lib_dir is None:
            lib_dir = DEFAULT_LIB_DIR
        self.combos = self.get_combos()
        if render_mode is None:
            render_human = False
        elif render_mode == "rgb_array":
            render_human = True
        else:
            raise Exception(f"invalid render mode {render_mode}")
        if rand_seed is None:
            rand_seed = create_random_seed()
        options.update({
            "env_name": env_name,
            "num_levels": num_levels,
            "start_level": start_level,
            "num_actions": len(self.combos),
            "use_sequential_levels": bool(use_sequential_levels),
            "debug_mode": debug_mode,
            "rand_seed": rand_seed,
            "num_threads": num_threads,
            "render_human": render_human,
            "resource_root": resource_root,
        })
        if extra_options:
            # extension options understood only by the HIP library (include/procgen_amd.h)
            options.update(extra_options)
        super().__init__(lib_dir=lib_dir, num=num, options=options, buffer_padding=buffer_padding)

    def get_state(self):
        """reference procgen/env.py:138-146, one entry per env.  Libraries with the batched hook (procgen_amd_get_states,
        include/procgen_amd.h) are asked for 256 envs at a time; any other libenv.so (the compiled reference) per env."""
        import ctypes as C

        if not hasattr(self._lib, "procgen_amd_get_states"):
            buf = C.create_string_buffer(MAX_STATE_SIZE)
            result = []
            for env_idx in range(self.num):
                n = self.call_c_func("get_state", env_idx, buf, MAX_STATE_SIZE)
                result.append(C.string_at(buf, n))
            return result
        fn = self._lib.procgen_amd_get_states
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p]
        block = 256
        cap = 32 * MAX_STATE_SIZE
        buf = np.empty(cap, dtype=np.uint8)
        offs = np.zeros(block + 1, dtype=np.int64)
        result = []
        first = 0
        while first < self.num:
            want = min(block, self.num - first)
            got = fn(self._handle, first, want, buf.ctypes.data, cap, offs.ctypes.data)
            assert got >= 1
            mv = memoryview(buf)
            result.extend(bytes(mv[offs[k]:offs[k + 1]]) for k in range(got))
            first += got
        return result

    def set_state(self, states):
        """reference procgen/env.py:148-153.  Libraries with the batched hook (procgen_amd_set_states, include/procgen_amd.h) take 256 states
        per call -- one upload and one redraw per block; any other libenv.so (the compiled reference) is restored env by env."""
        assert len(states) == self.num
        if not hasattr(self._lib, "procgen_amd_set_states") or self.num < 4:
            for env_idx in range(self.num):
                state = states[env_idx]
                self.call_c_func("set_state", env_idx, state, len(state))
            return
        import ctypes as C

        fn = self._lib.procgen_amd_set_states
        fn.restype = None
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_void_p]
        block = 256
        for first in range(0, self.num, block):
            chunk = states[first:first + block]
            offs = np.zeros(len(chunk) + 1, dtype=np.int64)
            np.cumsum([len(st) for st in chunk], out=offs[1:])
            fn(self._handle, first, len(chunk), b"".join(chunk), offs.ctypes.data)

    def get_combos(self):
        return [("LEFT", "DOWN"), ("LEFT",), ("LEFT", "UP"), ("DOWN",), (), ("UP",), ("RIGHT", "DOWN"),
                ("RIGHT",), ("RIGHT", "UP"), ("D",), ("A",), ("W",), ("S",), ("Q",), ("E",)]

    def act(self, ac):
        return super().act({"action": np.asarray(ac).astype(np.int32)})


class ProcgenGym3Env(BaseProcgenEnv):
    """reference procgen/env.py:203-246"""

    def __init__(self, num, env_name, center_agent=True, use_backgrounds=True, use_monochrome_assets=False,
                 restrict_themes=False, use_generated_assets=False, paint_vel_info=False,
                 distribution_mode="hard", **kwargs):
        assert distribution_mode in DISTRIBUTION_MODE_DICT, f'"{distribution_mode}" is not a valid distribution mode.'
        if distribution_mode == "exploration":
            assert env_name in EXPLORATION_LEVEL_SEEDS, f"{env_name} does not support exploration mode"
            distribution_mode = DISTRIBUTION_MODE_DICT["hard"]
            assert "num_levels" not in kwargs, "exploration mode overrides num_levels"
            kwargs["num_levels"] = 1
            assert "start_level" not in kwargs, "exploration mode overrides start_level"
            kwargs["start_level"] = EXPLORATION_LEVEL_SEEDS[env_name]
        else:
            distribution_mode = DISTRIBUTION_MODE_DICT[distribution_mode]
        options = {
            "center_agent": bool(center_agent),
            "use_generated_assets": bool(use_generated_assets),
            "use_monochrome_assets": bool(use_monochrome_assets),
            "restrict_themes": bool(restrict_themes),
            "use_backgrounds": bool(use_backgrounds),
            "paint_vel_info": bool(paint_vel_info),
            "distribution_mode": distribution_mode,
        }
        super().__init__(num, env_name, options, **kwargs)
