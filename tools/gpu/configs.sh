# one-GPU shares of BASELINE configs[2..4]: bigfish 65536, starpilot 32768 (= 262144 / 8), all 16 games 16384 (= 131072 / 8)
python bench.py --game bigfish --steps 120 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
python bench.py --game starpilot --num-envs 32768 --steps 120 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
python bench.py --game all16 --num-envs 16384 --steps 120 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-600
python bench.py --game all16 --num-envs 65536 --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
python bench.py --host-landed --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
