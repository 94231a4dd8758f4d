# round 6, call 30: the idle time between two steps (80 us of 1160 in profiles/r06_final_timeline.txt).  (1) PROCGEN_AMD_ORDER=4: chunk 0's step
# grid on the main stream, in order behind the upload of the actions, instead of behind a cross-queue event; (2) PROCGEN_AMD_SPIN_US: libenv_observe
# polls the stream for the end of the frame kernels instead of blocking on it.  ab_bench.py, M steps/s, two repetitions each
TAG=${1:-r6c30}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
for g in coinrun bigfish starpilot bossfight; do
for o in 0 4; do for s in 0 3000; do
  echo -n "$g ORDER=$o SPIN_US=$s  "; PROCGEN_AMD_ORDER=$o PROCGEN_AMD_SPIN_US=$s timeout 200 python tools/gpu/ab_bench.py procgen_amd/csrc/build $g 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo
done; done; done | tee gpurun_out/${TAG}_order_spin.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -n 4 -k "launch_shape or golden_rollout" 2>&1 | tail -3 | tee gpurun_out/${TAG}_pytest.log
