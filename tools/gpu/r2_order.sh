R=$GRAFT_REPO_ROOT
cd $R
b() { python bench.py --steps 100 --warmup 20 --no-cpu-baseline --game $1 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value']/1e6,2), j['ms_per_step'], j['roofline']['kernel_ms_per_step'])"; }
for g in coinrun starpilot; do
for cfg in "0 2" "1 2" "2 2" "1 3" "1 4" "2 3" "0 3"; do set -- $cfg; echo -n "$g order $1 chunks $2: "; PROCGEN_AMD_ORDER=$1 PROCGEN_AMD_CHUNKS=$2 b $g; done; done
