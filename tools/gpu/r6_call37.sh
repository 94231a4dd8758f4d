# round 6, call 37: the pull form's set-up looks only at cells in columns / rows that have a pixel, and a frame without a grid cell on screen
# skips the rest of the set-up and the grid pass (fruitbot: only out-of-bounds wall columns beside the screen; dodgeball: a world of SPACE);
# bossfight's rotation pool at 48 records.  Whole GPU parity file, same-box A/B against the build before (build_prev) for all 16 games
TAG=${1:-r6c37}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_render_human.py -q -m gpu -x -n 4 -k "not protocol_at_its_own_length" 2>&1 | tail -3 | tee gpurun_out/${TAG}_pytest.log
timeout 1500 python tools/gpu/ab_bench.py procgen_amd/csrc/build_prev,procgen_amd/csrc/build fruitbot,dodgeball,bossfight,coinrun,maze,miner,chaser,climber,heist,ninja,jumper,leaper,caveflyer,bigfish,starpilot,plunder 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_ab.txt
