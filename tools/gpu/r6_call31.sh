# round 6, call 31: (1) PROCGEN_AMD_ORDER=4 (chunk 0 on the main stream) against the default order for all 16 games; (2) the upload of the actions by a
# shader copy instead of an SDMA engine (HSA_ENABLE_SDMA=0); (3) host time inside libenv_act / libenv_observe at 65 536 envs
TAG=${1:-r6c31}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
for s in 1 0; do echo -n "coinrun HSA_ENABLE_SDMA=$s  "; HSA_ENABLE_SDMA=$s timeout 200 python tools/gpu/ab_bench.py procgen_amd/csrc/build coinrun 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo; done | tee gpurun_out/${TAG}_sdma.txt
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_host_timing.txt
import sys, time
import numpy as np
sys.path.insert(0, ".")
from procgen_amd import ProcgenGym3Env
n = 65536
env = ProcgenGym3Env(n, "coinrun", rand_seed=23, extra_options={"host_observations": False})
acts = np.random.RandomState(0).randint(0, 15, size=(320, n), dtype=np.int32)
env.observe()
ta = to = tl = 0.0
t_prev = None
for t in range(320):
    t0 = time.perf_counter(); env.act(acts[t]); t1 = time.perf_counter(); env.observe(); t2 = time.perf_counter()
    if t >= 20:
        ta += t1 - t0; to += t2 - t1
        if t_prev is not None: tl += t0 - t_prev
    t_prev = t2
print(f"coinrun n={n}: libenv_act {ta / 300 * 1e6:.1f} us  libenv_observe {to / 300 * 1e6:.1f} us  python between observe and act {tl / 300 * 1e6:.1f} us  (per step)")
PY
for g in coinrun bigfish maze climber miner starpilot fruitbot leaper plunder heist ninja dodgeball bossfight chaser caveflyer jumper; do
for o in 0 4; do
  echo -n "$g ORDER=$o  "; PROCGEN_AMD_ORDER=$o timeout 200 python tools/gpu/ab_bench.py procgen_amd/csrc/build $g 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo
done; done | tee gpurun_out/${TAG}_order16.txt
