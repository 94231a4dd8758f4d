"""PROCGEN_AMD_RENDER_ORDER (round 5: the counting sort runs on the device, kernels.hip render_order_*) on one MI355X: (1) the frames of an ordered handle equal those of a default one, (2) what the order buys.

(1) 4096 and 12288 envs (two launch chunks), 150 steps with the order rebuilt every 16: observation CRCs, rewards and firsts against a
    default handle stepped with the same actions.
(2) per game, 65536 envs, after WARM steps: device ms per step (procgen_amd_time_steps: HIP events around exactly what libenv_act
    enqueues) with the order off and with K in KS; the host's rebuild happens outside the bracket, its cost is printed apart
    (wall-clock steps/s over a rebuild period).

usage: python tools/gpu/render_order_ab.py [games=coinrun,climber,ninja,jumper,bigfish,starpilot] [N=65536]
"""
import ctypes as C
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from procgen_amd import ProcgenGym3Env  # noqa: E402

WARM, KS = 1200, (16, 64)


def make(game, n, k):
    if k:
        os.environ["PROCGEN_AMD_RENDER_ORDER"] = str(k)
    else:
        os.environ.pop("PROCGEN_AMD_RENDER_ORDER", None)
    env = ProcgenGym3Env(n, game, rand_seed=23, extra_options={"host_observations": True})
    env.observe()
    return env


def same_frames(game, n, steps=150, k=16):
    acts = np.random.RandomState(1).randint(0, 15, size=(steps, n), dtype=np.int32)
    out = []
    for kk in (0, k):
        env = make(game, n, kk)
        crc, rews, firsts = [], 0.0, 0
        for a in acts:
            env.act(a)
            rew, ob, first = env.observe()
            crc.append(zlib.crc32(ob["rgb"].tobytes()))
            rews += float(rew.sum())
            firsts += int(first.sum())
        env.close()
        out.append((crc, rews, firsts))
    ok = out[0] == out[1]
    print(f"same frames {game} N={n}: {'OK' if ok else 'MISMATCH'} (episodes ended: {out[0][2]})", flush=True)
    return ok


def timing(game, n):
    acts = np.random.RandomState(0).randint(0, 15, size=(WARM, n), dtype=np.int32)
    kacts = np.ascontiguousarray(np.random.RandomState(2).randint(0, 15, size=(64, n), dtype=np.int32))
    row = []
    for k in (0,) + KS:
        if k:
            os.environ["PROCGEN_AMD_RENDER_ORDER"] = str(k)
        else:
            os.environ.pop("PROCGEN_AMD_RENDER_ORDER", None)
        env = ProcgenGym3Env(n, game, rand_seed=23, extra_options={"host_observations": False})
        env._lib.procgen_amd_time_steps.restype = C.c_double
        env._lib.procgen_amd_time_steps.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        env.observe()
        for a in acts:
            env.act(a)
            env.observe()
        ms = env._lib.procgen_amd_time_steps(env._handle, 64, kacts.ctypes.data)
        period = max(k, 64)
        t0 = time.perf_counter()
        for t in range(period):
            env.act(acts[t])
            env.observe()
        wall = n * period / (time.perf_counter() - t0)
        env.close()
        row.append(f"K={k}: {ms:.4f} ms/step on the device, {wall / 1e6:.2f} M steps/s wall over {period} steps")
    print(game, " | ".join(row), flush=True)


if __name__ == "__main__":
    games = (sys.argv[1] if len(sys.argv) > 1 else "coinrun,climber,ninja,jumper,bigfish,starpilot").split(",")
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    good = all([same_frames(g, m) for g in games[:2] for m in (4096, 12288)])
    if not good:
        sys.exit("the ordered launch changed frames: do not time it")
    for g in games:
        timing(g, n)
