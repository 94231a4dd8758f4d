# round 4: phase tables of the four slowest games (current build)
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
for g in fruitbot bossfight dodgeball leaper; do echo "=== $g"; PROCGEN_AMD_DEBUG=2048 python bench.py --game $g --steps 60 --warmup 10 --no-cpu-baseline --steady-warmup 0 2>&1 | grep -v amdgpu.ids | grep -v "^{" | head -30; done | tee gpurun_out/r4_phase_games.txt
