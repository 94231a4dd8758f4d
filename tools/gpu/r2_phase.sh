R=$GRAFT_REPO_ROOT
cd $R
for cfg in "32 4" "16 2" "8 1"; do set -- $cfg
echo "== phase: 1 chunk, limits $1 / $2"; PROCGEN_AMD_LANE_ENTS=$1 PROCGEN_AMD_LANE_SMART=$2 PROCGEN_AMD_CHUNKS=1 PROCGEN_AMD_DEBUG=2048 python bench.py --steps 60 --warmup 20 --no-cpu-baseline 2>gpurun_out/r2_phase_$1.txt | tail -1 | cut -c1-100
grep -A 16 "lane = env kernel" gpurun_out/r2_phase_$1.txt
done
