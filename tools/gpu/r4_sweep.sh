# round 4: the long-horizon full-size parity test, then launch-shape re-sweep on the new kernels and the config shares
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity_at_scale.py -q -k "long_horizon" 2>&1 | tail -3 | tee gpurun_out/r4_sweep_pytest.log
b() { python bench.py --steps 150 --warmup 20 --no-cpu-baseline --steady-warmup 0 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M steps/s', d['ms_per_step'], 'ms')"; }
{
echo "chunks=1: $(PROCGEN_AMD_CHUNKS=1 b)"
echo "chunks=2 first 75 (default): $(b)"
echo "chunks=2 first 60: $(PROCGEN_AMD_FIRST_PCT=60 b)"
echo "chunks=2 first 85: $(PROCGEN_AMD_FIRST_PCT=85 b)"
echo "chunks=3: $(PROCGEN_AMD_CHUNKS=3 b)"
echo "bigfish 65536: $(b --game bigfish)"
echo "starpilot 32768: $(b --game starpilot --num-envs 32768)"
echo "all16 joint 16384: $(b --game all16 --num-envs 16384)"
} | tee gpurun_out/r4_sweep.txt
