# small / joint handles: parity of the single-stream launch path and its rates
python -m pytest tests/test_gpu_parity.py -x -q -k "joint or seeding or option_surface or overflow_routing" 2>&1 | tail -2
for a in "--game all16 --num-envs 16384 --steps 120 --warmup 20" "--game all16 --num-envs 65536 --steps 60 --warmup 10" "--game coinrun --num-envs 64 --steps 300 --warmup 20" "--game coinrun --num-envs 2048 --steps 300 --warmup 20"; do
python bench.py $a --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$a', d['value'], d['ms_per_step'])"; done
