# round 6, call 38: the per-cell on-screen check of build_pull_tables only for the games that ask for it (GRID_RARELY_ON_SCREEN: fruitbot, dodgeball):
# same-box A/B against the build of the second checkpoint (build_prev), then the closing checkpoint script once more on this library
TAG=${1:-r6_final4}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout 900 python tools/gpu/ab_bench.py procgen_amd/csrc/build_prev,procgen_amd/csrc/build fruitbot,dodgeball,bossfight,coinrun,ninja,caveflyer,jumper,climber 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_ab_vs_checkpoint2.txt
bash tools/gpu/r6_final.sh ${TAG}
