"""Same-box A/B of two libenv.so builds (steps/s of the default bench loop, device-resident observations)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from procgen_amd import ProcgenGym3Env

def run(lib, game, n, steps=100, warm=20):
    # the atlas directory is resolved relative to the library: point both at the repo's
    os.environ["PROCGEN_AMD_DATA_DIR"] = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "procgen_amd", "data")
    env = ProcgenGym3Env(n, game, rand_seed=23, lib_dir=lib, extra_options={"host_observations": False})
    acts = np.random.RandomState(0).randint(0, 15, size=(warm + steps, n), dtype=np.int32)
    env.observe()
    for t in range(warm):
        env.act(acts[t]); env.observe()
    t0 = time.perf_counter()
    for t in range(warm, warm + steps):
        env.act(acts[t]); env.observe()
    dt = time.perf_counter() - t0
    env.close()
    return n * steps / dt

if __name__ == "__main__":
    libs = sys.argv[1].split(",")
    games = sys.argv[2].split(",")
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
    for game in games:
        for rep in range(2):
            print(game, " ".join(f"{os.path.basename(l)}={run(l, game, n) / 1e6:.2f}M" for l in libs), flush=True)
