R=$GRAFT_REPO_ROOT
cd $R
b() { python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step'])"; }
echo "default (lane off, tile 1)"; b
echo "lane on 16/1"; PROCGEN_AMD_LANE=1 b
echo "lane on, nothing routed"; PROCGEN_AMD_LANE=1 PROCGEN_AMD_LANE_ENTS=0 b
python -m pytest tests/test_gpu_parity.py -x -q -k "(coinrun and (golden or parity or forced or state)) or entity_table or arena_tiers" 2>&1 | tail -2
PROCGEN_AMD_LANE=1 python -m pytest tests/test_gpu_parity.py -x -q -k "(coinrun and (golden or parity or forced or state)) or entity_table" 2>&1 | tail -2
