# round 6, call 42: the serial suite uncaptured (-s), so that whatever the aborting process writes to stderr is kept
TAG=${1:-r6c42}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export PROCGEN_AMD_FATAL_LOG=$R/gpurun_out/${TAG}_fatal.log
for i in 1 2; do
  ( time timeout 1500 python -X faulthandler -m pytest tests -x -v -s -m gpu ) > gpurun_out/${TAG}_pytest_full_$i.log 2>&1
  tail -3 gpurun_out/${TAG}_pytest_full_$i.log | cut -c1-200
  if ! grep -q ' passed' gpurun_out/${TAG}_pytest_full_$i.log; then echo "RUN $i DID NOT FINISH"; grep -n -B25 'Fatal Python' gpurun_out/${TAG}_pytest_full_$i.log | grep -v amdgpu.ids | cut -c1-300 | tail -40; break; fi
done
