# round 6, call 41: the abort inside get_state of test_set_state_leaves_one_list_entry_whatever_the_tier_order (2 of 5 serial suite runs): the test
# alone and behind its predecessors, uncaptured (-s: pytest's fd capture swallows what the dying process wrote to stderr), until it dies
TAG=${1:-r6c41}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export PROCGEN_AMD_FATAL_LOG=$R/gpurun_out/${TAG}_fatal.log
for i in $(seq 1 12); do
  timeout 300 python -X faulthandler -m pytest -s -x -q -m gpu "tests/test_gpu_parity_at_scale.py::test_set_state_leaves_one_list_entry_whatever_the_tier_order" > gpurun_out/${TAG}_alone_$i.log 2>&1
  rc=$?; echo "alone $i rc=$rc"; if [ $rc -ne 0 ]; then grep -v '^  File' gpurun_out/${TAG}_alone_$i.log | tail -30 | cut -c1-300; break; fi; rm -f gpurun_out/${TAG}_alone_$i.log
done
for i in $(seq 1 6); do
  timeout 900 python -X faulthandler -m pytest -s -x -q -m gpu tests/test_gpu_parity_at_scale.py -k "timeout_classes or separately_placed or set_state_leaves" > gpurun_out/${TAG}_seq_$i.log 2>&1
  rc=$?; echo "sequence $i rc=$rc"; if [ $rc -ne 0 ]; then grep -v '^  File' gpurun_out/${TAG}_seq_$i.log | tail -40 | cut -c1-300; break; fi; rm -f gpurun_out/${TAG}_seq_$i.log
done
