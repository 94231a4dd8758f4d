# round 6, call 16: envs per prep wave (2 / 4 / 8), the release build (-DPG_RELEASE) against the default build on all 16 games, bigfish's launch order by background
TAG=${1:-r6c16}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout 600 python tools/gpu/ab_bench.py procgen_amd/csrc/build,procgen_amd/csrc/build_p2,procgen_amd/csrc/build_p8 coinrun 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_prep_envs_ab.txt
timeout 1800 python tools/gpu/ab_bench.py procgen_amd/csrc/build,procgen_amd/csrc/build_rel coinrun,bigfish,maze,climber,miner,starpilot,fruitbot,leaper,plunder,heist,ninja,dodgeball,bossfight,chaser,caveflyer,jumper 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_release_ab.txt
timeout 600 python tools/gpu/render_order_ab.py bigfish,coinrun 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_render_order_ab.txt
