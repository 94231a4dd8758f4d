# round 3 quick check: parity subset, then bench lines (new path / A-B switch) and per-phase cycles
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_parity.py -x -q -n 4 -k "coinrun or climber or jumper or dodgeball or modes or option" 2>&1 | tail -3
for g in coinrun climber bigfish starpilot jumper leaper; do
  a=$(python bench.py --game $g --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['value']/1e6,2))")
  b=$(PROCGEN_AMD_DEBUG=32768 python bench.py --game $g --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['value']/1e6,2))")
  echo "$g: new $a M steps/s, A/B switch (old path) $b"
done
PROCGEN_AMD_DEBUG=2048 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | grep -B16 "render kernel" | head -16
