# step/render overlap: how many env chunks (PROCGEN_AMD_CHUNKS) the default workload wants
for c in 1 2 3 4 6; do PROCGEN_AMD_CHUNKS=$c python bench.py --steps 150 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('chunks $c', d['value'], d['ms_per_step'])"; done
