# round 6, call 11: coinrun's arena without size-class tables (6.5 KB: six waves per SIMD if the registers allow): with the full renderer inside raster (96 VGPRs) and without (59)
TAG=${1:-r6c11}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -n 4 -k "display_list or (coinrun and not protocol_at_its_own)" 2>&1 | tail -3 | tee gpurun_out/${TAG}_pytest_dl.log
PROCGEN_AMD_FIRST_PCT=60 timeout 600 python tools/gpu/ab_bench.py procgen_amd/csrc/build_r05,procgen_amd/csrc/build,procgen_amd/csrc/build_ns coinrun 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_ab.txt
cd /tmp && export TMPDIR=/tmp
export PROCGEN_AMD_LIB_DIR=$R/procgen_amd/csrc/build_ns
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --kernel-trace -d /tmp/${TAG}_f0 -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-host-landed --no-traffic > $R/gpurun_out/${TAG}_f0.log 2>&1
python $R/tests/tools/rocpd_summary.py $(find /tmp/${TAG}_f0 -name "*.db" | head -1) > $R/gpurun_out/${TAG}_f0.csv 2>&1
grep "raster\|4prepI" $R/gpurun_out/${TAG}_f0.csv | cut -c1-130 | grep -v "SQ_"
