# round 3 checkpoint on one MI355X: smoke, whole GPU suite (4 workers), the r3 "after" profile of the default workload, all 16 games
R=$GRAFT_REPO_ROOT
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python -m pytest tests -m gpu -q -n 4 2>&1 | tail -5
bash tools/gpu/r3_profile.sh ${1:-r3_after}
cd $R
bash tools/gpu/bench16.sh 2>&1 | tee gpurun_out/${1:-r3_after}_bench16.log | tail -20
