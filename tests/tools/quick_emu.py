"""Quick bit-exactness loop while iterating on the kernels (CPU): the emulated kernel sources against the plain-C oracle.
  PG_EMU_GAMES=CoinRun python tests/tools/quick_emu.py coinrun [envs] [steps] [key=value options ...]
(PG_EMU_GAMES builds tests/emu for that subset only: seconds instead of minutes.)"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests"), os.path.join(REPO, "tests", "emu")):
    sys.path.insert(0, p)
import numpy as np
import emu_harness, oracle_env
from helpers import action_stream

game = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
kw = {}
for a in sys.argv[4:]:
    k, v = a.split("=")
    kw[k] = int(v) if v.lstrip("-").isdigit() else (v == "True" if v in ("True", "False") else v)
t0 = time.time()
acts = action_stream(n, steps, seed=5)
orc = oracle_env.OracleEnv(n, game, rand_seed=41, **kw)
emu = emu_harness.EmuEnv(n, game, rand_seed=41, **kw)
bad = 0
for t in range(steps + 1):
    r1, o1, f1 = orc.observe()
    r2, o2, f2 = emu.observe()
    if not (np.array_equal(r1, r2) and np.array_equal(f1, f2)):
        print(f"step {t}: rew/first differ"); bad += 1
    if not np.array_equal(o1["rgb"], o2["rgb"]):
        d = np.abs(o1["rgb"].astype(int) - o2["rgb"].astype(int)).max(axis=3)
        envs = [int(e) for e in np.nonzero(d.reshape(n, -1).max(axis=1))[0]]
        e = envs[0]
        ys, xs = np.nonzero(d[e])
        print(f"step {t}: frames differ in envs {envs[:8]}; env {e}: {len(ys)} px, rows {ys.min()}-{ys.max()}, cols {xs.min()}-{xs.max()}"); bad += 1
    if bad > 5:
        break
    if t < steps:
        orc.act(acts[t]); emu.act(acts[t])
import ctypes
emu.L.emu_counter.restype = ctypes.c_longlong
c7 = emu.L.emu_counter(7)
if c7:
    print(f"rotation-record pool: {c7 // 1000000} frames sent to the per-band path, {c7 % 1000000} windows cut")
emu.L.emu_frame_count.restype = ctypes.c_longlong
print("display list: frames drawn from their record", emu.L.emu_frame_count(emu.h, 0), "/ by the full renderer", emu.L.emu_frame_count(emu.h, 1))
print(f"{game} {kw}: {n} envs x {steps} steps, {bad} mismatching steps, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
