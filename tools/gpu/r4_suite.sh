# round 4: the whole GPU suite, then the same-box A/B against the round-3 library for the three BASELINE games
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -12 | tee gpurun_out/r4_suite_pytest.log
python tools/gpu/ab_bench.py tools/gpu/ab/libenv_r03.so,procgen_amd/csrc/build/libenv.so coinrun,bigfish,starpilot 2>&1 | tee gpurun_out/r4_suite_ab.txt
