# round 6, call 33: the step's actions uploaded by a copy kernel on the step's stream (PROCGEN_AMD_ACTION_COPY=1, 16 bytes per lane from the pinned
# host buffer) instead of hipMemcpyAsync (an SDMA engine whose completion the step's first kernels wait for); build_ak, ab_bench.py, M steps/s
TAG=${1:-r6c33}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
for g in coinrun bigfish starpilot bossfight; do
for a in 0 1 0 1; do
  echo -n "$g ACTION_COPY=$a  "; PROCGEN_AMD_ACTION_COPY=$a timeout 200 python tools/gpu/ab_bench.py procgen_amd/csrc/build_ak $g 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo
done; done | tee gpurun_out/${TAG}_action_copy.txt
PROCGEN_AMD_ACTION_COPY=1 PROCGEN_AMD_LIB_DIR=$R/procgen_amd/csrc/build_ak timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -n 4 -k "golden_rollout or parity_with_oracle" 2>&1 | tail -3 | tee gpurun_out/${TAG}_pytest.log
