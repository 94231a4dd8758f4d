"""Where a small handle's step time goes: host time inside libenv_act (launch calls) vs libenv_observe (wait + copies)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from procgen_amd import ProcgenGym3Env
import bench

for game, n in (("coinrun", 64), ("coinrun", 2048), (",".join(bench.ALL_GAMES), 16384)):
    env = ProcgenGym3Env(n, game, rand_seed=23, extra_options={"host_observations": False})
    rng = np.random.RandomState(0)
    env.observe()
    ta = to = 0.0
    for t in range(220):
        a = rng.randint(0, 15, size=(n,), dtype=np.int32)
        t0 = time.perf_counter(); env.act(a); t1 = time.perf_counter(); env.observe(); t2 = time.perf_counter()
        if t >= 20:
            ta += t1 - t0; to += t2 - t1
    print(f"{game[:20]:20s} n={n:6d}  act {ta / 200 * 1e3:.3f} ms  observe {to / 200 * 1e3:.3f} ms", flush=True)
    env.close()
