# lane = env path: coinrun parity subset, bench, kernel trace
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_parity.py -x -q -k "(coinrun and (golden or parity or forced)) or entity_table or arena_tiers" 2>&1 | tail -3 | tee gpurun_out/r2_lane_pytest.log
python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2_lane_bench.json; cut -c1-200 gpurun_out/r2_lane_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt -o kt -- python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $R/gpurun_out/r2_lane_kt.log 2>&1
python $R/tests/tools/rocpd_summary.py $(find $R/gpurun_out/kt -name "*.db" | head -1) > $R/gpurun_out/r2_lane_kernel_trace.csv 2>&1
rm -rf $R/gpurun_out/kt
head -8 $R/gpurun_out/r2_lane_kernel_trace.csv
