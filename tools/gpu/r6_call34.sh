# round 6, call 34: where the step time of the two slowest games goes: kernel trace + per-dispatch timeline of bossfight and fruitbot (steady state)
TAG=${1:-r6c34}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for g in bossfight fruitbot; do
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_$g -o kt -- python $R/bench.py --game $g --steps 48 --warmup 5 --no-cpu-baseline --no-host-landed --no-traffic > $R/gpurun_out/${TAG}_$g.log 2>&1
  DB=$(find /tmp/${TAG}_$g -name "*.db" | head -1)
  python $R/tests/tools/rocpd_summary.py $DB > $R/gpurun_out/${TAG}_${g}_kernel_trace.csv 2>&1
  python $R/tests/tools/rocpd_timeline.py $DB > $R/gpurun_out/${TAG}_${g}_timeline.txt 2>&1
  rm -rf /tmp/${TAG}_$g
  echo "== $g"; head -12 $R/gpurun_out/${TAG}_${g}_kernel_trace.csv | cut -c1-150; head -40 $R/gpurun_out/${TAG}_${g}_timeline.txt | tail -24 | cut -c1-110
done
