# round 4: whole GPU suite on the new kernels (every lane section now reads its lane id through an opaque asm), then a same-box A/B against the
# round-3 library (tools/gpu/ab/libenv_r03.so, not tracked) and the phase table
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -n 4 2>&1 | tail -6 | tee gpurun_out/r4_ab_pytest.log
python tools/gpu/ab_bench.py tools/gpu/ab/libenv_r03.so,procgen_amd/csrc/build/libenv.so coinrun,bigfish,starpilot,fruitbot,leaper,dodgeball,bossfight,maze 2>&1 | tee gpurun_out/r4_ab_bench.txt
PROCGEN_AMD_DEBUG=2048 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | grep -A40 "phase cycles" | head -30 | tee gpurun_out/r4_ab_phase.txt
