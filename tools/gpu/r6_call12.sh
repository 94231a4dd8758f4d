# round 6, call 12: display list v4 (slow frames on a list kernel again, raster at six waves per SIMD for coinrun): whole GPU suite, A/B per game and first-chunk share, bench line
TAG=${1:-r6c12}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export PROCGEN_AMD_FATAL_LOG=$R/gpurun_out/${TAG}_fatal.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/${TAG}_smoke.txt
timeout 1200 python -m pytest tests -q -m gpu -n 4 -k "not protocol_at_its_own_length" 2>&1 | tail -12 > gpurun_out/${TAG}_pytest.log; tail -5 gpurun_out/${TAG}_pytest.log
for p in 60 75; do echo "FIRST_PCT=$p"; PROCGEN_AMD_FIRST_PCT=$p timeout 900 python tools/gpu/ab_bench.py procgen_amd/csrc/build_r05,procgen_amd/csrc/build coinrun,bigfish,maze,miner,climber,chaser 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/${TAG}_ab.txt
python bench.py --no-cpu-baseline 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-330 gpurun_out/${TAG}_bench.json
PROCGEN_AMD_FIRST_PCT=60 python bench.py --no-cpu-baseline --no-traffic --no-host-landed 2>/dev/null | tail -1 | cut -c1-330
( time timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -n 4 -k "protocol_at_its_own_length" ) 2>&1 | tail -6 | tee gpurun_out/${TAG}_pytest_state_protocol.log
