PROCGEN_AMD_DEBUG=2048 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | grep -A30 "render kernel" | head -14
