# round 6, call 17: multi-size rows_pass (maze, miner, climber, heist, jumper ...) and the per-game release flags: A/B against the round-5 library on all 16 games, parity subset, TCC counters for the render-order question
TAG=${1:-r6c17}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export PROCGEN_AMD_FATAL_LOG=$R/gpurun_out/${TAG}_fatal.log
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -n 4 -k "golden_rollout or parity_with_oracle or display_list or lds_dma or option_surface or distribution_mode" 2>&1 | tail -4 | tee gpurun_out/${TAG}_pytest.log
timeout 2400 python tools/gpu/ab_bench.py procgen_amd/csrc/build_r05,procgen_amd/csrc/build coinrun,bigfish,maze,climber,miner,starpilot,fruitbot,leaper,plunder,heist,ninja,dodgeball,bossfight,chaser,caveflyer,jumper 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_ab16.txt
bash tools/gpu/r6_tcc.sh ${TAG}_tcc 2>&1 | tail -30
