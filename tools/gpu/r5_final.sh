# round 5, closing checkpoint, most important first: the whole GPU suite serially (as the driver runs it), the default bench line with
# cpu_baseline, steady-state kernel trace + per-dispatch timeline, HBM-traffic PMC passes (-> profiles/r05_hbm_traffic.json), per-phase
# cycles, the 16 games alone, the joint handle's one-GPU share, the other config shares.
# usage: bash tools/gpu/r5_final.sh [tag]
TAG=${1:-r5_final}
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/${TAG}_smoke.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/${TAG}_pytest.log
python bench.py 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-600 gpurun_out/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_kt -o kt -- python $R/bench.py --steps 64 --warmup 5 --no-cpu-baseline > $R/gpurun_out/${TAG}_kt.log 2>&1
DB=$(find /tmp/${TAG}_kt -name "*.db" | head -1)
python $R/tests/tools/rocpd_summary.py $DB > $R/gpurun_out/${TAG}_kernel_trace.csv 2>&1
python $R/tests/tools/rocpd_timeline.py $DB > $R/gpurun_out/${TAG}_timeline.txt 2>&1
rm -rf /tmp/${TAG}_kt; head -8 $R/gpurun_out/${TAG}_kernel_trace.csv | cut -c1-170
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/${TAG}_pmc_$n -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${TAG}_pmc_$n.log 2>&1
  python $R/tests/tools/rocpd_summary.py $(find /tmp/${TAG}_pmc_$n -name "*.db" | head -1) > $R/gpurun_out/${TAG}_pmc_$n.csv 2>&1
  rm -rf /tmp/${TAG}_pmc_$n
done
cd $R
python tools/gpu/make_traffic_json.py ${TAG} gpurun_out/${TAG}_hbm_traffic.json 65536 2>&1 | tail -2
PROCGEN_AMD_DEBUG=2048 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | grep -v "^{" > gpurun_out/${TAG}_phase_cycles.txt; tail -14 gpurun_out/${TAG}_phase_cycles.txt
for g in coinrun bigfish maze climber miner starpilot fruitbot leaper plunder heist ninja dodgeball bossfight chaser caveflyer jumper; do python bench.py --game $g --steps 120 --warmup 20 --no-cpu-baseline --steady-warmup 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['config']['workload'].split()[0], round(d['value']/1e6,2))"; done 2>&1 | tee gpurun_out/${TAG}_bench16.log
python bench.py --game all16 --num-envs 16384 --steps 100 --warmup 20 --no-cpu-baseline --steady-warmup 0 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_all16_joint_16384.json; cut -c1-220 gpurun_out/${TAG}_bench_all16_joint_16384.json
python bench.py --game bigfish --steps 150 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_bigfish_65536.json; cut -c1-200 gpurun_out/${TAG}_bench_bigfish_65536.json
python bench.py --game starpilot --num-envs 32768 --steps 150 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_starpilot_32768.json; cut -c1-200 gpurun_out/${TAG}_bench_starpilot_32768.json
# one more run of the suite with four workers sharing the GPU (the eleventh of the round), fatal log kept
export PROCGEN_AMD_FATAL_LOG=$R/gpurun_out/${TAG}_fatal.log
timeout 900 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -4 | tee gpurun_out/${TAG}_pytest_parallel.log
grep "fatal:" $PROCGEN_AMD_FATAL_LOG | cut -c1-150 | sed 's/\[pid [0-9]*\] //' | sort | uniq -c
