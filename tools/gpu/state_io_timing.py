"""Wall time of env.get_state() / env.set_state() over a whole vector (the reference's state_test protocol, procgen/state_test.py:71-124,
at BASELINE sizes): python tools/gpu/state_io_timing.py [game=coinrun] [N=65536]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from procgen_amd import ProcgenGym3Env

game = sys.argv[1] if len(sys.argv) > 1 else "coinrun"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
env = ProcgenGym3Env(n, game, rand_seed=23)
rng = np.random.RandomState(0)
for _ in range(20):
    env.act(rng.randint(0, 15, size=(n,), dtype=np.int32))
env.observe()
t0 = time.perf_counter()
st = env.get_state()
t1 = time.perf_counter()
print(f"{game} N={n}: get_state of every env {t1 - t0:.2f} s ({sum(len(s) for s in st) / 1e6:.0f} MB, {n / (t1 - t0):.0f} states/s)", flush=True)
m = min(n, 4096)
t0 = time.perf_counter()
for e in range(m):
    env.call_c_func("set_state", e, st[(e + 1) % n], len(st[(e + 1) % n]))
t1 = time.perf_counter()
print(f"{game} N={n}: set_state of {m} envs {t1 - t0:.2f} s ({m / (t1 - t0):.0f} states/s)", flush=True)
got = env.get_state()[:m]
assert got == [st[(e + 1) % n] for e in range(m)], "states restored at other indices must come back byte for byte (game_n is adopted, reference src/game.cpp:253)"
# the whole vector through the batched hook (procgen_amd_set_states: 256 states per call, one upload + one redraw per block)
rot = st[1:] + st[:1]
t0 = time.perf_counter()
env.set_state(rot)
t1 = time.perf_counter()
print(f"{game} N={n}: batched set_state of every env {t1 - t0:.2f} s ({n / (t1 - t0):.0f} states/s)", flush=True)
assert env.get_state() == rot
env.close()
