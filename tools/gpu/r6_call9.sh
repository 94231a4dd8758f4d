# round 6, call 9: the row-major grid pass (Renderer::rows_pass) against the cell-row-major one (PROCGEN_AMD_DEBUG & 2097152), new GPU tests
TAG=${1:-r6c9}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -n 4 -k "display_list or (coinrun and not protocol_at_its_own) or batched_set_states" 2>&1 | tail -3 | tee gpurun_out/${TAG}_pytest_dl.log
PROCGEN_AMD_FIRST_PCT=60 timeout 300 python tools/gpu/ab_bench.py procgen_amd/csrc/build_r05,procgen_amd/csrc/build coinrun 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_ab.txt
cd /tmp && export TMPDIR=/tmp
for f in 0 2 2097152; do
  PROCGEN_AMD_DEBUG=$f timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --kernel-trace -d /tmp/${TAG}_f$f -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-host-landed --no-traffic > $R/gpurun_out/${TAG}_f$f.log 2>&1
  python $R/tests/tools/rocpd_summary.py $(find /tmp/${TAG}_f$f -name "*.db" | head -1) > $R/gpurun_out/${TAG}_f$f.csv 2>&1
  rm -rf /tmp/${TAG}_f$f
  echo "== debug $f"; grep "raster" $R/gpurun_out/${TAG}_f$f.csv | cut -c1-130
done
cd $R
timeout 900 python tools/gpu/state_io_timing.py 2>&1 | tail -12 | tee gpurun_out/${TAG}_state_io.txt
