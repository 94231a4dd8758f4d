# round 6, call 44: the serial suite under pytest's default fd capture (the mode both aborts came under) with tools/gpu/abort_bt preloaded: if it
# aborts, the C backtrace of the raising thread is in gpurun_out/<tag>_abort_bt.txt
TAG=${1:-r6c44}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export PROCGEN_AMD_FATAL_LOG=$R/gpurun_out/${TAG}_fatal.log
export ABORT_BT_FILE=$R/gpurun_out/${TAG}_abort_bt.txt
( time LD_PRELOAD=$R/tools/gpu/abort_bt/libabort_bt.so timeout 1100 python -m pytest tests -x -q -m gpu ) > gpurun_out/${TAG}_pytest.log 2>&1
tail -5 gpurun_out/${TAG}_pytest.log | cut -c1-200
[ -f $ABORT_BT_FILE ] && { echo "ABORT BACKTRACE:"; head -60 $ABORT_BT_FILE | cut -c1-200; }
