# round 3 final checkpoint on one MI355X, most important first (the call may be cut by the GPU budget):
#   smoke, whole GPU suite (4 workers), default bench line (with cpu_baseline), kernel trace + timeline, HBM traffic PMC passes,
#   per-phase wave cycles, instruction mix, the 16 games alone, the 16-game joint handle (profiles/r03_final_* come from this script)
# usage: bash tools/gpu/r3_final.sh [tag]
TAG=${1:-r3_final}
R=$GRAFT_REPO_ROOT
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python -m pytest tests -m gpu -q -n 4 2>&1 | tail -5 | tee gpurun_out/${TAG}_pytest.log
python bench.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-400 gpurun_out/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_kt -o kt -- python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $R/gpurun_out/${TAG}_kt.log 2>&1
DB=$(find $R/gpurun_out/${TAG}_kt -name "*.db" | head -1)
python $R/tests/tools/rocpd_summary.py $DB > $R/gpurun_out/${TAG}_kernel_trace.csv 2>&1
python $R/tests/tools/rocpd_timeline.py $DB 3 > $R/gpurun_out/${TAG}_timeline.txt 2>&1
rm -rf $R/gpurun_out/${TAG}_kt
head -8 $R/gpurun_out/${TAG}_kernel_trace.csv
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/${TAG}_pmc_$n -o p -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${TAG}_pmc_$n.log 2>&1
  python $R/tests/tools/rocpd_summary.py $(find $R/gpurun_out/${TAG}_pmc_$n -name "*.db" | head -1) > $R/gpurun_out/${TAG}_pmc_$n.csv 2>&1
  rm -rf $R/gpurun_out/${TAG}_pmc_$n/
done
grep -h "render\|step_" $R/gpurun_out/${TAG}_pmc_*.csv | grep -v "^_ZN.*kd,[0-9]*,[0-9.]*," | cut -c1-200 | head -24
cd $R
PROCGEN_AMD_DEBUG=2048 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | grep -B16 -A14 "render kernel" > gpurun_out/${TAG}_phase_cycles.txt; cat gpurun_out/${TAG}_phase_cycles.txt
bash tools/gpu/bench16.sh 2>&1 | tee gpurun_out/${TAG}_bench16.log
python bench.py --game all16 --num-envs 16384 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_all16_joint_16384.json; cut -c1-300 gpurun_out/${TAG}_bench_all16_joint_16384.json
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/${TAG}_pmc_$n -o p -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${TAG}_pmc_$n.log 2>&1
  python $R/tests/tools/rocpd_summary.py $(find $R/gpurun_out/${TAG}_pmc_$n -name "*.db" | head -1) > $R/gpurun_out/${TAG}_pmc_$n.csv 2>&1
  rm -rf $R/gpurun_out/${TAG}_pmc_$n/
done
