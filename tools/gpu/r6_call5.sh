# round 6, call 5: occupancy of the display-list kernels (prep arena 4 KB; raster with 16- / 8-row bands, with / without the full renderer inside)
# usage: bash tools/gpu/r6_call5.sh [tag]
TAG=${1:-r6c5}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/${TAG}_smoke.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -n 4 -k "display_list or coinrun" 2>&1 | tail -3 | tee gpurun_out/${TAG}_pytest_dl.log
LIBS=""
for v in build_r05 build build_b8 build_b16f build_b8f; do [ -f procgen_amd/csrc/$v/libenv.so ] && LIBS=$LIBS,procgen_amd/csrc/$v; done
timeout 900 python tools/gpu/ab_bench.py ${LIBS#,} coinrun 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_ab.txt
cd /tmp && export TMPDIR=/tmp
for v in build build_b8f; do
  export PROCGEN_AMD_LIB_DIR=$R/procgen_amd/csrc/$v
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_kt_$v -o kt -- python $R/bench.py --steps 64 --warmup 5 --no-cpu-baseline --no-host-landed --no-traffic > $R/gpurun_out/${TAG}_kt_$v.log 2>&1
  python $R/tests/tools/rocpd_summary.py $(find /tmp/${TAG}_kt_$v -name "*.db" | head -1) > $R/gpurun_out/${TAG}_kernel_trace_$v.csv 2>&1
  python $R/tests/tools/rocpd_timeline.py $(find /tmp/${TAG}_kt_$v -name "*.db" | head -1) > $R/gpurun_out/${TAG}_timeline_$v.txt 2>&1
  echo "== $v"; tail -1 $R/gpurun_out/${TAG}_kt_$v.log | cut -c1-200; head -7 $R/gpurun_out/${TAG}_kernel_trace_$v.csv | cut -c1-150; sed -n 20,34p $R/gpurun_out/${TAG}_timeline_$v.txt
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d /tmp/${TAG}_f0_$v -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-host-landed --no-traffic > $R/gpurun_out/${TAG}_f0_$v.log 2>&1
  python $R/tests/tools/rocpd_summary.py $(find /tmp/${TAG}_f0_$v -name "*.db" | head -1) > $R/gpurun_out/${TAG}_f0_$v.csv 2>&1
  grep "raster\|4prep" $R/gpurun_out/${TAG}_f0_$v.csv | cut -c1-160
done
