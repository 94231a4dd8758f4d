# round 6, call 43: the serial suite under --capture=sys (Python-level capture only: what C code writes to fd 2 reaches the log), until it aborts
TAG=${1:-r6c43}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export PROCGEN_AMD_FATAL_LOG=$R/gpurun_out/${TAG}_fatal.log
for i in 1 2; do
  ( time timeout 1200 python -X faulthandler -m pytest tests -x -v --capture=sys -m gpu ) > gpurun_out/${TAG}_pytest_full_$i.log 2>&1
  tail -3 gpurun_out/${TAG}_pytest_full_$i.log | cut -c1-200
  if ! grep -q ' passed' gpurun_out/${TAG}_pytest_full_$i.log; then echo "RUN $i DID NOT FINISH"; grep -v 'amdgpu.ids' gpurun_out/${TAG}_pytest_full_$i.log | grep -n -B30 'Fatal Python' | cut -c1-300 | tail -45; break; fi
done
