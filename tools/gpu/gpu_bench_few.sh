for g in coinrun ninja climber jumper caveflyer dodgeball leaper; do python bench.py --game $g --steps 120 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['config']['workload'], d['value'])"; done
PROCGEN_AMD_DEBUG=2048 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | grep -i -A40 "phase" | head -60
