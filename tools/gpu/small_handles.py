"""Round 4: a step of a small handle (the micro-benchmark shape of reference procgen/env_test.py:55-68: 1 / 2 / 16 envs, and the sizes up
to the point where the kernels take over), through the plain ABI with observations landed in the caller's array.
   python tools/gpu/small_handles.py            (PROCGEN_AMD_NO_ZEROCOPY=1: the copy-based step of round 3)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from procgen_amd import ProcgenGym3Env

for game, n in (("coinrun", 1), ("coinrun", 2), ("coinrun", 16), ("coinrun", 64), ("coinrun", 255), ("coinrun", 1024), ("coinrun", 2048), ("bigfish", 64), ("starpilot", 64)):
    env = ProcgenGym3Env(n, game, rand_seed=23)
    rng = np.random.RandomState(0)
    env.observe()
    ta = to = 0.0
    steps = 400
    for t in range(steps + 40):
        a = rng.randint(0, 15, size=(n,), dtype=np.int32)
        t0 = time.perf_counter(); env.act(a); t1 = time.perf_counter(); env.observe(); t2 = time.perf_counter()
        if t >= 40:
            ta += t1 - t0; to += t2 - t1
    ms = (ta + to) / steps * 1e3
    print(f"{game:10s} n={n:5d}  step {ms:.3f} ms (act {ta / steps * 1e3:.3f} + observe {to / steps * 1e3:.3f})  {n / ms / 1e3:.3f} M steps/s", flush=True)
    env.close()
