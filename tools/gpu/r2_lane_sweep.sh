# The lane = env experiment: coinrun 65536 envs with PROCGEN_AMD_LANE=1, routing bounds swept, and the
# lane kernel's per-phase wave cycles (PROCGEN_AMD_DEBUG=2048) -> profiles/r02_lane_phase_cycles.txt
R=$GRAFT_REPO_ROOT
cd $R
export PROCGEN_AMD_LANE=1
b() { python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step'])"; }
python -m pytest tests/test_gpu_parity.py -x -q -k "(coinrun and (golden or parity or forced or state)) or entity_table" 2>&1 | tail -2
echo "lane off"; PROCGEN_AMD_LANE=0 b
for cfg in "8 1" "16 1" "16 2" "32 4"; do set -- $cfg; echo "bounds $1 / $2"; PROCGEN_AMD_LANE_ENTS=$1 PROCGEN_AMD_LANE_SMART=$2 b; done
for cfg in "32 4" "16 2" "8 1"; do set -- $cfg
echo "== phase cycles, one chunk, bounds $1 / $2"; PROCGEN_AMD_LANE_ENTS=$1 PROCGEN_AMD_LANE_SMART=$2 PROCGEN_AMD_CHUNKS=1 PROCGEN_AMD_DEBUG=2048 python bench.py --steps 60 --warmup 20 --no-cpu-baseline 2>gpurun_out/r2_phase_$1.txt | tail -1 | cut -c1-100
grep -A 16 "lane = env kernel" gpurun_out/r2_phase_$1.txt
done
