# round 5, fifth GPU call (short): why the 16-game joint handle's one-GPU share came out below round 4's (16.2 vs 17.2 M): issuing threads
# (round 5 raised the pool from min(parts, 8) to min(parts, cores, 32)) x render launch order on the 1024-env parts; then coinrun's per-phase
# cycles with the launch order off / on
TAG=${1:-r5c5}
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
J="python bench.py --game all16 --num-envs 16384 --steps 150 --warmup 20 --no-cpu-baseline --steady-warmup 0"
for th in 16 8 4; do for ro in default 0; do
  if [ $ro = default ]; then unset PROCGEN_AMD_RENDER_ORDER; else export PROCGEN_AMD_RENDER_ORDER=$ro; fi
  PROCGEN_AMD_HOST_THREADS=$th $J 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('host_threads=$th render_order=$ro', round(d['value']/1e6,2), 'M', d['ms_per_step'], 'ms/step')"
done; done 2>&1 | tee gpurun_out/${TAG}_joint_ab.txt
unset PROCGEN_AMD_RENDER_ORDER
for ro in 0 16; do
  echo "--- PROCGEN_AMD_RENDER_ORDER=$ro"
  PROCGEN_AMD_RENDER_ORDER=$ro PROCGEN_AMD_DEBUG=2048 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | grep -v "^{" | tail -13
done 2>&1 | tee gpurun_out/${TAG}_phase_order.txt
