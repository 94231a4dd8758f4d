# round 6, call 40: hunting the abort of call 38's serial suite: the serial suite twice more, verbose, everything kept; stop at the first abort
TAG=${1:-r6c40}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export PROCGEN_AMD_FATAL_LOG=$R/gpurun_out/${TAG}_fatal.log
for i in 1 2; do
  ( time timeout 2400 python -X faulthandler -m pytest tests -x -v -m gpu ) > gpurun_out/${TAG}_pytest_full_$i.log 2>&1
  rc=$?
  tail -4 gpurun_out/${TAG}_pytest_full_$i.log | cut -c1-200
  if ! grep -q ' passed' gpurun_out/${TAG}_pytest_full_$i.log; then echo "RUN $i DID NOT FINISH"; grep -n 'Fatal Python\|Aborted\|Memory access fault\|HSA\|hip\|free()\|terminate\|Current thread\|File "' gpurun_out/${TAG}_pytest_full_$i.log | head -40 | cut -c1-250; break; fi
done
