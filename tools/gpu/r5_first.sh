# first GPU call of round 5, most important first (the call may be cut short):
#   smoke + default bench line; the launch-order experiment of the end of round 4 (PROCGEN_AMD_RENDER_ORDER, never run on a device):
#   same frames, device ms per step off / on, the render kernel's duration and FETCH_SIZE off / on; then the whole GPU suite with four
#   workers three times (do the null-stream joins hold? DESIGN.md section 5) and once serially.
# usage: bash tools/gpu/r5_first.sh [tag]
TAG=${1:-r5_first}
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-300 gpurun_out/${TAG}_bench.json
timeout 900 python tools/gpu/render_order_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_render_order_ab.txt
cd /tmp && export TMPDIR=/tmp
for k in 0 64; do
  export PROCGEN_AMD_RENDER_ORDER=$k; [ $k = 0 ] && unset PROCGEN_AMD_RENDER_ORDER
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_kt$k -o kt -- python $R/bench.py --steps 64 --warmup 1500 --no-cpu-baseline --steady-warmup 0 > $R/gpurun_out/${TAG}_kt$k.log 2>&1
  python $R/tests/tools/rocpd_summary.py $(find $R/gpurun_out/${TAG}_kt$k -name "*.db" | head -1) > $R/gpurun_out/${TAG}_kernel_trace_order$k.csv 2>&1
  rm -rf $R/gpurun_out/${TAG}_kt$k
  head -5 $R/gpurun_out/${TAG}_kernel_trace_order$k.csv
  for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $c | cut -d' ' -f1)
    rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/${TAG}_pmc_${n}_$k -o p -- python $R/bench.py --steps 8 --warmup 1500 --no-cpu-baseline --steady-warmup 0 > $R/gpurun_out/${TAG}_pmc_${n}_$k.log 2>&1
    python $R/tests/tools/rocpd_summary.py $(find $R/gpurun_out/${TAG}_pmc_${n}_$k -name "*.db" | head -1) > $R/gpurun_out/${TAG}_pmc_${n}_order$k.csv 2>&1
    rm -rf $R/gpurun_out/${TAG}_pmc_${n}_$k/
  done
done
unset PROCGEN_AMD_RENDER_ORDER
grep -h "render" $R/gpurun_out/${TAG}_pmc_*.csv | cut -c1-200 | head -20
cd $R
for i in 1 2 3; do
  PROCGEN_AMD_FATAL_LOG=$R/gpurun_out/${TAG}_fatal.log timeout 900 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -6 | tee gpurun_out/${TAG}_pytest_parallel_$i.log
done
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/${TAG}_pytest.log
