# round 5, fourth GPU call (short): SourceOver only on the rows of the grid pass where a translucent cell image really lands (pg_render.h
# draw_tiles_pull stage 1) -- build_blend -- against the default build, same box, the games that draw their grid in pull form
TAG=${1:-r5c4}
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
PROCGEN_AMD_LIB_DIR=$R/procgen_amd/csrc/build_blend timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 4 -k "oracle or fixture" 2>&1 | tail -3 | tee gpurun_out/${TAG}_parity_build_blend.log
timeout 900 python tools/gpu/ab_bench.py procgen_amd/csrc/build,procgen_amd/csrc/build_blend coinrun,climber,ninja,maze,miner,heist,chaser,jumper,caveflyer 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_blend_ab.txt
