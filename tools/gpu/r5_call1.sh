# round 5, first GPU call, most important first (the call may be cut short):
#   1 smoke + bench line (steady-state `value`, events in the timed loop)
#   2 the never-run variants of the end of round 4: PROCGEN_AMD_RENDER_ORDER (same frames; device ms / step, render duration, FETCH_SIZE and
#     TCC hit rate off / on), the rotation-record pool builds (parity subset, then the same-box A/B)
#   3 the whole GPU suite with four workers, N times, fatal log kept (does the rare device-side check still fire under GPU sharing?)
#   4 a PC-sampling run of the steady-state bench (stochastic, else host trap), summarised per instruction
# usage: bash tools/gpu/r5_call1.sh [tag] [suite runs]
TAG=${1:-r5c1}
RUNS=${2:-6}
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
export PROCGEN_AMD_FATAL_LOG=$R/gpurun_out/${TAG}_fatal.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/${TAG}_smoke.txt
python bench.py --no-cpu-baseline 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-1200 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err
# ---- 2a launch order by background image
timeout 600 python tools/gpu/render_order_ab.py coinrun,climber,ninja,bigfish 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_render_order_ab.txt
cd /tmp && export TMPDIR=/tmp
for k in 0 64; do
  export PROCGEN_AMD_RENDER_ORDER=$k; [ $k = 0 ] && unset PROCGEN_AMD_RENDER_ORDER
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_kt$k -o kt -- python $R/bench.py --steps 64 --warmup 5 --no-cpu-baseline > $R/gpurun_out/${TAG}_kt$k.log 2>&1
  python $R/tests/tools/rocpd_summary.py $(find $R/gpurun_out/${TAG}_kt$k -name "*.db" | head -1) > $R/gpurun_out/${TAG}_kernel_trace_order$k.csv 2>&1
  rm -rf $R/gpurun_out/${TAG}_kt$k
  head -5 $R/gpurun_out/${TAG}_kernel_trace_order$k.csv | cut -c1-160
  for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $c | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/${TAG}_pmc_${n}_$k -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${TAG}_pmc_${n}_$k.log 2>&1
    python $R/tests/tools/rocpd_summary.py $(find $R/gpurun_out/${TAG}_pmc_${n}_$k -name "*.db" | head -1) > $R/gpurun_out/${TAG}_pmc_${n}_order$k.csv 2>&1
    rm -rf $R/gpurun_out/${TAG}_pmc_${n}_$k/
  done
done
unset PROCGEN_AMD_RENDER_ORDER
grep -h "render" $R/gpurun_out/${TAG}_pmc_*.csv | cut -c1-200 | head -12
cd $R
# ---- 2b rotation-record pool builds
LIBS=procgen_amd/csrc/build
for v in build_pool16 build_pool16w4; do
  [ -f procgen_amd/csrc/$v/libenv.so ] || { echo "missing procgen_amd/csrc/$v/libenv.so"; continue; }
  LIBS=$LIBS,procgen_amd/csrc/$v
  PROCGEN_AMD_LIB_DIR=$R/procgen_amd/csrc/$v timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 4 -k "oracle or fixture" 2>&1 | tail -3 | tee gpurun_out/${TAG}_parity_$v.log
done
timeout 900 python tools/gpu/ab_bench.py $LIBS heist,caveflyer,plunder,starpilot,dodgeball,leaper,fruitbot,bossfight,jumper 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_pool_ab.txt
# same-box A/B of the default build against the round-4 library (what the error record, the late error check and the events cost)
[ -f procgen_amd/csrc/build_r04/libenv.so ] && timeout 300 python tools/gpu/ab_bench.py procgen_amd/csrc/build_r04,procgen_amd/csrc/build coinrun,bigfish 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_ab_vs_r04.txt
# ---- 3 the suite under GPU sharing
for i in $(seq 1 $RUNS); do
  timeout 900 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -8 > gpurun_out/${TAG}_pytest_parallel_$i.log
  tail -2 gpurun_out/${TAG}_pytest_parallel_$i.log
done
[ -f $PROCGEN_AMD_FATAL_LOG ] && { echo "--- fatal log"; grep -v "use_generated_assets\|another distribution_mode" $PROCGEN_AMD_FATAL_LOG | cut -c1-3000; }
# ---- 4 PC sampling (beta; under its own timeout)
cd /tmp
for m in stochastic host_trap; do
  u=cycles; iv=1048576; [ $m = host_trap ] && { u=time; iv=100; }
  timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $m --pc-sampling-unit $u --pc-sampling-interval $iv --kernel-trace --output-format csv -d /tmp/pcs_$m -o pcs -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/${TAG}_pcs_$m.log 2>&1
  echo "pc sampling $m rc=$?"; tail -3 $R/gpurun_out/${TAG}_pcs_$m.log | cut -c1-300
  python $R/tests/tools/pcsamp_summary.py /tmp/pcs_$m render 150 > $R/gpurun_out/${TAG}_pcsamp_${m}_render.txt 2>&1
  python $R/tests/tools/pcsamp_summary.py /tmp/pcs_$m step_tier0 80 > $R/gpurun_out/${TAG}_pcsamp_${m}_step.txt 2>&1
  head -40 $R/gpurun_out/${TAG}_pcsamp_${m}_render.txt | cut -c1-200
  grep -q "samples per kernel" $R/gpurun_out/${TAG}_pcsamp_${m}_render.txt && break
done
