# round 6, call 24: (1) LDS allocation granule of gfx950 (workgroups resident per CU against the static LDS size), (2) release flags per kernel
# family: the frame kernels of the games that are not built with -DPG_RELEASE compiled without the profiling apparatus (-DPG_RELEASE_FRAME)
TAG=${1:-r6c24}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout 120 tools/gpu/micro/residency_lds 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_residency_lds.txt
for g in coinrun maze miner chaser heist starpilot bossfight; do
  timeout 300 python tools/gpu/ab_bench.py procgen_amd/csrc/build,procgen_amd/csrc/build_relf $g 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/${TAG}_ab.txt
