# round 4: early download of the small outputs (behind the step kernels, scattered on the host while the frames are drawn): on / off, and parity
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity_at_scale.py tests/test_gpu_parity.py -x -q -n 4 2>&1 | tail -3 | tee gpurun_out/r4_early_pytest.log
for g in coinrun bigfish jumper; do
  a=$(python tools/gpu/ab_bench.py procgen_amd/csrc/build/libenv.so $g 2>&1 | tail -1)
  b=$(PROCGEN_AMD_EARLY_SMALL=0 python tools/gpu/ab_bench.py procgen_amd/csrc/build/libenv.so $g 2>&1 | tail -1)
  echo "$a | off: $b"
done | tee gpurun_out/r4_early.txt
