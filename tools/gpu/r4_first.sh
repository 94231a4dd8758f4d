# round 4, first call: (a) render_human x use_generated_assets on the device, (b) the sharded-handle sections in a loop, 4 processes sharing the GPU,
# PROCGEN_AMD_HOST_THREADS 1 and 8, (c) a reference bench line + phase cycles on this box
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
PROCGEN_AMD_GEN_RENDER_HUMAN=1 timeout 300 python -m pytest tests/test_render_human.py -m gpu -k gen -q 2>&1 | tail -15 | tee gpurun_out/r4_gen_human.log
for th in 1 8; do
  for p in 0 1 2 3; do
    PROCGEN_AMD_HOST_THREADS=$th timeout 500 python tools/gpu/shard_stress.py 8 "th$th.p$p" > gpurun_out/r4_stress_th${th}_p$p.log 2>&1 &
  done
  wait
  cat gpurun_out/r4_stress_th${th}_p*.log | tail -40
done
python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | tee gpurun_out/r4_first_bench.json
PROCGEN_AMD_DEBUG=2048 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | grep -A40 "phase cycles" | head -40 | tee gpurun_out/r4_first_phase.txt
