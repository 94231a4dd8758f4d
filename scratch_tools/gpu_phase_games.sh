for g in leaper bossfight maze miner fruitbot jumper; do echo "=== $g"; PROCGEN_AMD_DEBUG=2048 python bench.py --game $g --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | grep -v "^{" | head -32; done
