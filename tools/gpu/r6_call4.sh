# round 6, call 4: display list v2 (slow frames drawn inside prep, cell-row-major grid pass) on the device:
#   1 smoke, display-list tests, parity tests of the six display-list games
#   2 same-box A/B: round-5 library / one-kernel build of this round / prep without the occupancy hint / default, on coinrun; the other five display-list games r05 vs default
#   3 bench line, kernel trace + timeline, SQ instruction counters
# usage: bash tools/gpu/r6_call4.sh [tag]
TAG=${1:-r6c4}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export PROCGEN_AMD_FATAL_LOG=$R/gpurun_out/${TAG}_fatal.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/${TAG}_smoke.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -n 4 -k "display_list or coinrun or bigfish or maze or miner or climber or chaser or render_launch or lds_dma" 2>&1 | tail -12 | tee gpurun_out/${TAG}_pytest_dl.log
LIBS=""
for v in build_r05 build_opq build_nh build; do [ -f procgen_amd/csrc/$v/libenv.so ] && LIBS=$LIBS,procgen_amd/csrc/$v; done
timeout 600 python tools/gpu/ab_bench.py ${LIBS#,} coinrun 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_ab.txt
timeout 900 python tools/gpu/ab_bench.py procgen_amd/csrc/build_r05,procgen_amd/csrc/build bigfish,maze,miner,climber,chaser 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/${TAG}_ab.txt
python bench.py --no-cpu-baseline --no-traffic 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-400 gpurun_out/${TAG}_bench.json; tail -2 gpurun_out/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_kt -o kt -- python $R/bench.py --steps 64 --warmup 5 --no-cpu-baseline --no-host-landed --no-traffic > $R/gpurun_out/${TAG}_kt.log 2>&1
python $R/tests/tools/rocpd_summary.py $(find /tmp/${TAG}_kt -name "*.db" | head -1) > $R/gpurun_out/${TAG}_kernel_trace.csv 2>&1
python $R/tests/tools/rocpd_timeline.py $(find /tmp/${TAG}_kt -name "*.db" | head -1) > $R/gpurun_out/${TAG}_timeline.txt 2>&1
head -9 $R/gpurun_out/${TAG}_kernel_trace.csv | cut -c1-180; tail -14 $R/gpurun_out/${TAG}_timeline.txt
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES --kernel-trace -d /tmp/${TAG}_f0 -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-host-landed --no-traffic > $R/gpurun_out/${TAG}_f0.log 2>&1
python $R/tests/tools/rocpd_summary.py $(find /tmp/${TAG}_f0 -name "*.db" | head -1) > $R/gpurun_out/${TAG}_f0.csv 2>&1
grep "raster\|4prep\|step_tier0" $R/gpurun_out/${TAG}_f0.csv | cut -c1-200
