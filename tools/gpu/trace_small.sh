cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kts -o kt -- python $R/bench.py --num-envs 2048 --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/kts.log 2>&1
python $R/tests/tools/rocpd_summary.py $(find $R/gpurun_out/kts -name "*.db" | head -1) 2>&1 | head -12
rm -rf $R/gpurun_out/kts
