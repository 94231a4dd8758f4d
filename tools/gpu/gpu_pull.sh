python -m pytest tests/test_gpu_parity.py -x -q -k "parity_with_oracle_many_envs or golden_rollout" 2>&1 | tail -3
for g in maze miner leaper jumper climber ninja coinrun; do python bench.py --game $g --steps 120 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['config']['workload'][:12], d['value'])"; done
