R=$GRAFT_REPO_ROOT
cd $R
b() { python bench.py --steps 100 --warmup 20 --no-cpu-baseline --game $1 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value']/1e6,2), j['ms_per_step'])"; }
for g in coinrun bigfish starpilot maze; do for p in 0 20 25 30 35 65 70 75 0; do echo -n "$g first_pct $p: "; PROCGEN_AMD_FIRST_PCT=$p b $g; done; done
