# round 6, call 27: gridless games without column / row tables in the render arena (bossfight 11 632 -> 10 640 bytes: nine LDS granules, 14 frames per CU
# instead of 12), bossfight's tier-0 step arena at 120 entity slots instead of 128 (11 680 -> 11 008 bytes: the same step; build_bf), chaser's frame
# kernels without the profiling apparatus: same-box A/B against the build before (build_prev), tier counts of bossfight from bench.py
TAG=${1:-r6c27}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -n 4 -k "bossfight or bigfish or starpilot or plunder or chaser" 2>&1 | tail -3 | tee gpurun_out/${TAG}_pytest.log
timeout 900 python tools/gpu/ab_bench.py procgen_amd/csrc/build_prev,procgen_amd/csrc/build,procgen_amd/csrc/build_bf bossfight 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_ab.txt
timeout 900 python tools/gpu/ab_bench.py procgen_amd/csrc/build_prev,procgen_amd/csrc/build bigfish,starpilot,plunder,chaser 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/${TAG}_ab.txt
for b in build build_bf; do PROCGEN_AMD_LIB_DIR=$R/procgen_amd/csrc/$b python bench.py --game bossfight --no-cpu-baseline --no-traffic --no-host-landed 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_bossfight_$b.json; python -c "
import json,sys; j=json.load(open('gpurun_out/${TAG}_bench_bossfight_$b.json')); print('$b', j['value'], json.dumps(j)[:0], [ (k, v) for k, v in j.items() if isinstance(v, dict) and 'arena_tier_envs' in v ])"; done | tee gpurun_out/${TAG}_bossfight_tiers.txt
