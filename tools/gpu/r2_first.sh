# Round 2, first GPU contact of the lane = env path: coinrun parity subset, bench with the lane kernel on / off, kernel trace.
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_parity.py -x -q -k "coinrun or entity_table or arena_tiers or option_surface_matches or full_size or native" 2>&1 | tail -4 | tee gpurun_out/r2_first_pytest.log
python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2_first_bench_lane.json; cut -c1-260 gpurun_out/r2_first_bench_lane.json
PROCGEN_AMD_DEBUG=4096 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2_first_bench_nolane.json; cut -c1-260 gpurun_out/r2_first_bench_nolane.json
PROCGEN_AMD_CHUNKS=1 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2_first_bench_lane_1chunk.json; cut -c1-260 gpurun_out/r2_first_bench_lane_1chunk.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt -o kt -- python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $R/gpurun_out/r2_first_kt.log 2>&1
python $R/tests/tools/rocpd_summary.py $(find $R/gpurun_out/kt -name "*.db" | head -1) > $R/gpurun_out/r2_first_kernel_trace.csv 2>&1
rm -rf $R/gpurun_out/kt
head -12 $R/gpurun_out/r2_first_kernel_trace.csv
