# round 3: refresh the "where do the cycles go" evidence for the default workload (coinrun, 65536 envs):
#   bench line, per-phase wave cycles, kernel trace (+ a per-dispatch timeline of three steps), instruction-mix PMC
# usage: bash tools/gpu/r3_profile.sh <tag>
TAG=${1:-r3}
R=$GRAFT_REPO_ROOT
cd $R
python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-260 gpurun_out/${TAG}_bench.json
PROCGEN_AMD_DEBUG=2048 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | grep -B16 -A14 "render kernel" > gpurun_out/${TAG}_phase_cycles.txt; cat gpurun_out/${TAG}_phase_cycles.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_kt -o kt -- python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $R/gpurun_out/${TAG}_kt.log 2>&1
DB=$(find $R/gpurun_out/${TAG}_kt -name "*.db" | head -1)
python $R/tests/tools/rocpd_summary.py $DB > $R/gpurun_out/${TAG}_kernel_trace.csv 2>&1
python $R/tests/tools/rocpd_timeline.py $DB 3 > $R/gpurun_out/${TAG}_timeline.txt 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/${TAG}_pmc_$n -o p -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${TAG}_pmc_$n.log 2>&1
  python $R/tests/tools/rocpd_summary.py $(find $R/gpurun_out/${TAG}_pmc_$n -name "*.db" | head -1) > $R/gpurun_out/${TAG}_pmc_$n.csv 2>&1
done
rm -rf $R/gpurun_out/${TAG}_kt $R/gpurun_out/${TAG}_pmc_*/
head -8 $R/gpurun_out/${TAG}_kernel_trace.csv
head -40 $R/gpurun_out/${TAG}_timeline.txt
