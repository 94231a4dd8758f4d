# round 6, call 22: launch-order variants (PROCGEN_AMD_ORDER: where the tier-2 list kernel runs) x first-chunk share on the final kernels, coinrun
TAG=${1:-r6c22}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
for o in 0 1 2 3; do for p in 50 60 70; do
  echo -n "ORDER=$o FIRST_PCT=$p  "; PROCGEN_AMD_ORDER=$o PROCGEN_AMD_FIRST_PCT=$p timeout 200 python tools/gpu/ab_bench.py procgen_amd/csrc/build coinrun 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo
done; done | tee gpurun_out/${TAG}_order_sweep.txt
