# Round-end measurement on one MI355X: GPU test suite, the default bench line, the kernel trace and the HBM-traffic
# counters (each PMC set in its own pass, no trace domains besides --kernel-trace).  Summaries land in gpurun_out/.
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/final_pytest.log
python bench.py 2>/dev/null | tail -1 > gpurun_out/final_bench_coinrun.json; cat gpurun_out/final_bench_coinrun.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt -o kt -- python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $R/gpurun_out/final_kt.log 2>&1
python $R/tests/tools/rocpd_summary.py $(find $R/gpurun_out/kt -name "*.db" | head -1) > $R/gpurun_out/final_kernel_trace.csv 2>&1
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_$n -o p -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $R/gpurun_out/final_pmc_$n.log 2>&1
  python $R/tests/tools/rocpd_summary.py $(find $R/gpurun_out/pmc_$n -name "*.db" | head -1) > $R/gpurun_out/final_pmc_$n.csv 2>&1
done
rm -rf $R/gpurun_out/kt $R/gpurun_out/pmc_*/
head -8 $R/gpurun_out/final_kernel_trace.csv
grep -h "render\|step_tier0" $R/gpurun_out/final_pmc_*.csv | grep -v "^_ZN.*kd,[0-9]*,[0-9.]*," | head -12
