# Round-2 final measurements on one MI355X: the default bench line, per-game table (same-box A/B against the round-1
# library when tools/gpu/ab/libenv_r01.so is present), BASELINE config shares, kernel traces and the HBM-traffic counters
# (each PMC set in its own pass, no trace domains besides --kernel-trace).  Summaries land in gpurun_out/.
R=$GRAFT_REPO_ROOT
cd $R
python bench.py 2>/dev/null | tail -1 > gpurun_out/r02_bench_coinrun.json; cut -c1-200 gpurun_out/r02_bench_coinrun.json
for cfg in "bigfish 65536" "starpilot 32768" "all16 16384"; do set -- $cfg
  python bench.py --game $1 --num-envs $2 --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02_bench_$1_$2.json; cut -c1-40 gpurun_out/r02_bench_$1_$2.json | tr '\n' ' '; python -c "import json; j=json.load(open('gpurun_out/r02_bench_$1_$2.json')); print('$1', j['value'], j['ms_per_step'])"
done
LIBS=procgen_amd/csrc/build/libenv.so; [ -f tools/gpu/ab/libenv_r01.so ] && LIBS=tools/gpu/ab/libenv_r01.so,$LIBS
python tools/gpu/ab_bench.py $LIBS bigfish,bossfight,caveflyer,chaser,climber,coinrun,dodgeball,fruitbot,heist,jumper,leaper,maze,miner,ninja,plunder,starpilot 2>&1 | grep -v amdgpu.ids | awk 'NR%2==0' > gpurun_out/r02_all16_games.txt; cat gpurun_out/r02_all16_games.txt
python tools/gpu/host_timing.py 2>/dev/null | tail -4 > gpurun_out/r02_small_handles.txt; cat gpurun_out/r02_small_handles.txt
cd /tmp && export TMPDIR=/tmp
for g in coinrun bigfish starpilot; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt -o kt -- python $R/bench.py --game $g --steps 40 --warmup 10 --no-cpu-baseline > $R/gpurun_out/r02_kt_$g.log 2>&1
  python $R/tests/tools/rocpd_summary.py $(find $R/gpurun_out/kt -name "*.db" | head -1) > $R/gpurun_out/r02_kernel_trace_$g.csv 2>&1
  rm -rf $R/gpurun_out/kt; head -7 $R/gpurun_out/r02_kernel_trace_$g.csv
done
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_$n -o p -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r02_pmc_$n.log 2>&1
  python $R/tests/tools/rocpd_summary.py $(find $R/gpurun_out/pmc_$n -name "*.db" | head -1) > $R/gpurun_out/r02_pmc_$n.csv 2>&1
  rm -rf $R/gpurun_out/pmc_$n
done
grep -h "render\|step_tier0\|step_list" $R/gpurun_out/r02_pmc_*.csv | head -12
