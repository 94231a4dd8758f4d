# round 4: step kernel staging (one-trip load_env, one-round erase compaction): parity subset + A/B against the round-3 library + phase table
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_state_wire_format.py -m gpu -x -q -n 4 2>&1 | tail -3 | tee gpurun_out/r4_step_pytest.log
python tools/gpu/ab_bench.py tools/gpu/ab/libenv_r03.so,procgen_amd/csrc/build/libenv.so coinrun,bigfish,starpilot,maze,dodgeball,leaper,fruitbot,bossfight 2>&1 | tee gpurun_out/r4_step_ab.txt
PROCGEN_AMD_DEBUG=2048 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --steady-warmup 0 2>&1 | grep -A28 "phase cycles" | head -28 | tee gpurun_out/r4_step_phase.txt
