# round 6, call 7: where raster<CoinRun>'s time and instructions are (ablation bits 1 / 2 / 4 / 8 of PROCGEN_AMD_DEBUG in raster_env), prep alone (16 = no frame kernels at all is not it: 15 = raster set-up only)
TAG=${1:-r6c7}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for f in 0 1 2 4 8 15; do
  PROCGEN_AMD_DEBUG=$f timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --kernel-trace -d /tmp/${TAG}_f$f -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-host-landed --no-traffic > $R/gpurun_out/${TAG}_f$f.log 2>&1
  python $R/tests/tools/rocpd_summary.py $(find /tmp/${TAG}_f$f -name "*.db" | head -1) > $R/gpurun_out/${TAG}_f$f.csv 2>&1
  rm -rf /tmp/${TAG}_f$f
  echo "== debug $f"; grep "raster\|4prep" $R/gpurun_out/${TAG}_f$f.csv | cut -c1-130
done
