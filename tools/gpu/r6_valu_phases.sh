# round 6, call 1: where render<CoinRun>'s vector instructions are.  SQ_INSTS_VALU / SALU / LDS per render wave under the ablation bits of
# PROCGEN_AMD_DEBUG (1 no background, 2 no grid cells, 4 no entities, 8 no store; 15 = set-up alone): the differences are the phases' counts.
# usage: bash tools/gpu/r6_valu_phases.sh [tag]
TAG=${1:-r6_valu}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
python bench.py --no-cpu-baseline 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-600 gpurun_out/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
for f in 0 1 2 4 8 15; do
  PROCGEN_AMD_DEBUG=$f timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES --kernel-trace -d /tmp/${TAG}_f$f -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-host-landed > $R/gpurun_out/${TAG}_f$f.log 2>&1
  python $R/tests/tools/rocpd_summary.py $(find /tmp/${TAG}_f$f -name "*.db" | head -1) > $R/gpurun_out/${TAG}_f$f.csv 2>&1
  rm -rf /tmp/${TAG}_f$f
  echo "== debug $f"; grep "render" $R/gpurun_out/${TAG}_f$f.csv | cut -c1-200
done
