"""How many frames of a display-list game leave the rasterizer's short path at BASELINE size, and why (PROCGEN_AMD_DISPLAY_LIST_REPORT):
python tools/gpu/dl_probe.py [game=coinrun] [N=65536] [steps=200]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
os.environ["PROCGEN_AMD_DISPLAY_LIST_REPORT"] = "1"
from procgen_amd import ProcgenGym3Env

game = sys.argv[1] if len(sys.argv) > 1 else "coinrun"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
env = ProcgenGym3Env(n, game, rand_seed=23, extra_options={"host_observations": False})
rng = np.random.RandomState(0)
out = (C.c_int * 2)()
env._lib.procgen_amd_display_list_frames.restype = C.c_int
env.observe()
for t in range(steps):
    env.act(rng.randint(0, 15, size=(n,), dtype=np.int32))
    env.observe()
    if t % 20 == 19:
        env._lib.procgen_amd_display_list_frames(env._handle, out)
env.close()
