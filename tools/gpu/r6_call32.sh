# round 6, call 32: chunk 0's step grid submitted ahead of the list kernels (default now; PROCGEN_AMD_ORDER=5 = the submission order before),
# x chunk 0 on the main stream (ORDER=4), all 16 games; the timing events read one step late (bench.py line); switch tests
TAG=${1:-r6c32}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -n 4 -k "launch_shape or golden_rollout" 2>&1 | tail -3 | tee gpurun_out/${TAG}_pytest.log
for g in coinrun bigfish maze climber miner starpilot fruitbot leaper plunder heist ninja dodgeball bossfight chaser caveflyer jumper; do
for o in 5 0 4; do
  echo -n "$g ORDER=$o  "; PROCGEN_AMD_ORDER=$o timeout 200 python tools/gpu/ab_bench.py procgen_amd/csrc/build $g 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo
done; done | tee gpurun_out/${TAG}_order16.txt
python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-330 gpurun_out/${TAG}_bench.json
