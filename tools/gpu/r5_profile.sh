# round 5, profile-only call: the device-resident main loop alone (bench.py --no-host-landed: the host-landed leg's handle, four chunks under
# an 805 MB copy per step, would mix into every per-kernel average) -- kernel trace + timeline, the three HBM-traffic PMC passes, then the
# joint handle with the final issuing-thread default and the default bench line
TAG=${1:-r5_prof}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
B="python $R/bench.py --no-cpu-baseline --no-host-landed"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_kt -o kt -- $B --steps 64 --warmup 5 > $R/gpurun_out/${TAG}_kt.log 2>&1
DB=$(find /tmp/${TAG}_kt -name "*.db" | head -1)
python $R/tests/tools/rocpd_summary.py $DB > $R/gpurun_out/${TAG}_kernel_trace.csv 2>&1
python $R/tests/tools/rocpd_timeline.py $DB > $R/gpurun_out/${TAG}_timeline.txt 2>&1
rm -rf /tmp/${TAG}_kt; head -8 $R/gpurun_out/${TAG}_kernel_trace.csv | cut -c1-170; head -24 $R/gpurun_out/${TAG}_timeline.txt | cut -c1-120
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/${TAG}_pmc_$n -o p -- $B --steps 8 --warmup 2 > $R/gpurun_out/${TAG}_pmc_$n.log 2>&1
  python $R/tests/tools/rocpd_summary.py $(find /tmp/${TAG}_pmc_$n -name "*.db" | head -1) > $R/gpurun_out/${TAG}_pmc_$n.csv 2>&1
  rm -rf /tmp/${TAG}_pmc_$n
done
cd $R
python tools/gpu/make_traffic_json.py ${TAG} gpurun_out/${TAG}_hbm_traffic.json 65536 2>&1 | tail -2
python bench.py --game all16 --num-envs 16384 --steps 150 --warmup 20 --no-cpu-baseline --steady-warmup 0 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_all16_joint_16384.json; cut -c1-260 gpurun_out/${TAG}_bench_all16_joint_16384.json
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench.json; python -c "import json; d=json.load(open('gpurun_out/${TAG}_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'][:40], d['host_landed']['value'])"
