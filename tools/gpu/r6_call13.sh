# round 6, call 13: display list v5 (slow frames drawn by libenv_observe on a host-mapped flag: no list kernel behind raster; first chunk 60 %): tests, A/B, bench, 16 games
TAG=${1:-r6c13}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export PROCGEN_AMD_FATAL_LOG=$R/gpurun_out/${TAG}_fatal.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/${TAG}_smoke.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -n 4 -k "display_list or (coinrun and not protocol_at_its_own) or batched or launch_shape" 2>&1 | tail -4 | tee gpurun_out/${TAG}_pytest_dl.log
timeout 900 python tools/gpu/ab_bench.py procgen_amd/csrc/build_r05,procgen_amd/csrc/build coinrun,bigfish,maze,miner,climber,chaser 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_ab.txt
python bench.py --no-cpu-baseline 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-330 gpurun_out/${TAG}_bench.json
for g in bigfish bossfight caveflyer chaser climber coinrun dodgeball fruitbot heist jumper leaper maze miner ninja plunder starpilot; do
  timeout 200 python bench.py --game $g --steps 100 --warmup 20 --steady-warmup 0 --no-cpu-baseline --no-host-landed --no-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.readline()); print('$g', round(j['value'] / 1e6, 2), 'M steps/s', j['ms_per_step'], 'ms/step')"
done | tee gpurun_out/${TAG}_bench16.log
timeout 1200 python -m pytest tests -q -m gpu -n 4 -k "not protocol_at_its_own_length" 2>&1 | tail -6 > gpurun_out/${TAG}_pytest.log; tail -4 gpurun_out/${TAG}_pytest.log
