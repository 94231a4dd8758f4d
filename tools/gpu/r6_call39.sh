# round 6, call 39: the serial GPU suite of the closing checkpoint aborted (SIGABRT of the pytest process, 4 m 43 s in; the four-worker run of the same
# library passed): the same command again, verbose, everything kept
TAG=${1:-r6c39}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export PROCGEN_AMD_FATAL_LOG=$R/gpurun_out/${TAG}_fatal.log
( time timeout 2400 python -X faulthandler -m pytest tests -x -v -m gpu ) > gpurun_out/${TAG}_pytest_full.log 2>&1
tail -60 gpurun_out/${TAG}_pytest_full.log | cut -c1-300
dmesg 2>/dev/null | tail -20 | cut -c1-200
