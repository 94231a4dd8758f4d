R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_parity_at_scale.py -x -q 2>&1 | tail -5 | tee gpurun_out/r2_scale_pytest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt -o kt -- python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $R/gpurun_out/r2_default_kt.log 2>&1
python $R/tests/tools/rocpd_summary.py $(find $R/gpurun_out/kt -name "*.db" | head -1) > $R/gpurun_out/r2_default_kernel_trace.csv 2>&1
rm -rf $R/gpurun_out/kt
head -9 $R/gpurun_out/r2_default_kernel_trace.csv
