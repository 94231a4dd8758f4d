R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt -o kt -- python $R/bench.py --game all16 --num-envs 16384 --steps 40 --warmup 10 --no-cpu-baseline > $R/gpurun_out/r2_joint_kt.log 2>&1
python $R/tests/tools/rocpd_summary.py $(find $R/gpurun_out/kt -name "*.db" | head -1) > $R/gpurun_out/r02_joint_kernel_trace_final.csv 2>&1
rm -rf $R/gpurun_out/kt
cat $R/gpurun_out/r02_joint_kernel_trace_final.csv | cut -c1-130 | head -24
