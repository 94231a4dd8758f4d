# round 6, call 26: how many raster waves are resident?  SQ_WAVE_CYCLES x 4 / GRBM_GUI_ACTIVE per CU (clock-independent) for the build with the
# 5968-byte arena (build_prev), the 4688-byte one (build: 112 SGPRs -> seven waves per SIMD) and the same with amdgpu_waves_per_eu(8) (build_w8)
TAG=${1:-r6c26}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout 600 python tools/gpu/ab_bench.py procgen_amd/csrc/build_prev,procgen_amd/csrc/build,procgen_amd/csrc/build_w8 coinrun 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_ab.txt
cd /tmp && export TMPDIR=/tmp
for b in build_prev build build_w8; do
  PROCGEN_AMD_LIB_DIR=$R/procgen_amd/csrc/$b timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace -d /tmp/${TAG}_$b -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-host-landed --no-traffic > $R/gpurun_out/${TAG}_$b.log 2>&1
  python $R/tests/tools/rocpd_summary.py $(find /tmp/${TAG}_$b -name "*.db" | head -1) > $R/gpurun_out/${TAG}_$b.csv 2>&1
  rm -rf /tmp/${TAG}_$b
  echo "== $b"; grep -E '6rasterI|10step_tier0I|4prepI' $R/gpurun_out/${TAG}_$b.csv | grep -E 'kd,[0-9]+,[0-9.]+,|SQ_WAVES|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|GRBM_GUI_ACTIVE|INSTS'
done | tee $R/gpurun_out/${TAG}_residency.txt
