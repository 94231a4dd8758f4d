# round 4: small turned sprites in the groups of eight (exec_small_group): on / off (PROCGEN_AMD_DEBUG=524288) in the same build
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
for g in bossfight starpilot plunder caveflyer fruitbot heist dodgeball; do
  a=$(python tools/gpu/ab_bench.py procgen_amd/csrc/build/libenv.so $g 2>&1 | tail -1)
  b=$(PROCGEN_AMD_DEBUG=524288 python tools/gpu/ab_bench.py procgen_amd/csrc/build/libenv.so $g 2>&1 | tail -1)
  echo "$a | off: $b"
done | tee gpurun_out/r4_rot.txt
