R=$GRAFT_REPO_ROOT
cd $R
export PROCGEN_AMD_FATAL_LOG=$R/gpurun_out/r4_early_fatal.log
rm -f $PROCGEN_AMD_FATAL_LOG
timeout 900 python -m pytest tests/test_gpu_parity.py -x -v 2>&1 | grep -E "PASSED|FAILED|ERROR|passed|failed|^E " | cut -c1-200 | tail -25 | tee gpurun_out/r4_early_pytest2.log
cat $PROCGEN_AMD_FATAL_LOG 2>/dev/null | tail -5
