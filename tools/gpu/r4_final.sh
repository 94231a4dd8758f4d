# round 4 checkpoint on one MI355X, most important first (the call may be cut by the GPU budget):
#   smoke, default bench line (cpu_baseline + steady_state), kernel trace + timeline, HBM-traffic PMC passes and the instruction mix -- the
#   profiled runs in the STEADY STATE (1500 warm-up steps) --, per-phase wave cycles there, the 16 games alone, the 16-game joint handle,
#   the whole GPU suite serially (as the driver runs it).  profiles/r04_* come from this script.
# usage: bash tools/gpu/r4_final.sh [tag]
TAG=${1:-r4_final}
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-300 gpurun_out/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
WARM=1500
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_kt -o kt -- python $R/bench.py --steps 40 --warmup $WARM --no-cpu-baseline --steady-warmup 0 > $R/gpurun_out/${TAG}_kt.log 2>&1
DB=$(find $R/gpurun_out/${TAG}_kt -name "*.db" | head -1)
python $R/tests/tools/rocpd_summary.py $DB > $R/gpurun_out/${TAG}_kernel_trace.csv 2>&1
python $R/tests/tools/rocpd_timeline.py $DB 3 > $R/gpurun_out/${TAG}_timeline.txt 2>&1
rm -rf $R/gpurun_out/${TAG}_kt
head -8 $R/gpurun_out/${TAG}_kernel_trace.csv
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/${TAG}_pmc_$n -o p -- python $R/bench.py --steps 8 --warmup $WARM --no-cpu-baseline --steady-warmup 0 > $R/gpurun_out/${TAG}_pmc_$n.log 2>&1
  python $R/tests/tools/rocpd_summary.py $(find $R/gpurun_out/${TAG}_pmc_$n -name "*.db" | head -1) > $R/gpurun_out/${TAG}_pmc_$n.csv 2>&1
  rm -rf $R/gpurun_out/${TAG}_pmc_$n/
done
grep -h "render\|step_" $R/gpurun_out/${TAG}_pmc_*.csv | grep -v "^_ZN.*kd,[0-9]*,[0-9.]*," | cut -c1-200 | head -40
cd $R
PROCGEN_AMD_DEBUG=2048 python bench.py --steps 60 --warmup $WARM --no-cpu-baseline --steady-warmup 0 2>&1 | grep -B16 -A14 "render kernel" > gpurun_out/${TAG}_phase_cycles.txt; cat gpurun_out/${TAG}_phase_cycles.txt
for g in coinrun bigfish maze climber miner starpilot fruitbot leaper plunder heist ninja dodgeball bossfight chaser caveflyer jumper; do python bench.py --game $g --steps 120 --warmup 20 --no-cpu-baseline --steady-warmup 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['config']['workload'].split()[0], round(d['value']/1e6,2))"; done 2>&1 | tee gpurun_out/${TAG}_bench16.log
python bench.py --game all16 --num-envs 16384 --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_all16_joint_16384.json; cut -c1-200 gpurun_out/${TAG}_bench_all16_joint_16384.json
python bench.py --game bigfish --steps 150 --warmup 20 --no-cpu-baseline --steady-warmup 0 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_bigfish_65536.json
python bench.py --game starpilot --num-envs 32768 --steps 150 --warmup 20 --no-cpu-baseline --steady-warmup 0 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_starpilot_32768.json
python tools/gpu/small_handles.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_small_handles.txt
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/${TAG}_pytest.log
