# round 6, call 18: one draw routine for all command sets and layers (run_batch inlined twice in raster instead of seven times): tests, A/B, counters, bench
TAG=${1:-r6c18}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -n 4 -k "display_list or golden_rollout or (parity_with_oracle) or batched" 2>&1 | tail -3 | tee gpurun_out/${TAG}_pytest.log
timeout 900 python tools/gpu/ab_bench.py procgen_amd/csrc/build_r05,procgen_amd/csrc/build coinrun,bigfish,maze,miner,climber,chaser 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_ab.txt
python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-330 gpurun_out/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --kernel-trace -d /tmp/${TAG}_f0 -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-host-landed --no-traffic > $R/gpurun_out/${TAG}_f0.log 2>&1
python $R/tests/tools/rocpd_summary.py $(find /tmp/${TAG}_f0 -name "*.db" | head -1) > $R/gpurun_out/${TAG}_f0.csv 2>&1
grep "raster\|4prepI" $R/gpurun_out/${TAG}_f0.csv | cut -c1-130
