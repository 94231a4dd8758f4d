R=$GRAFT_REPO_ROOT
cd $R
b() { python bench.py --steps 100 --warmup 20 --no-cpu-baseline --game $1 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value']/1e6,2), j['ms_per_step'])"; }
for g in coinrun starpilot bigfish; do
echo -n "$g default 75/25: "; b $g
for cuts in "50,30,20" "60,25,15" "20,30,50" "45,35,20" "40,30,20,10" "55,25,12,8"; do n=$(echo $cuts | tr ',' '\n' | wc -l); echo -n "$g cuts $cuts: "; PROCGEN_AMD_CHUNKS=$n PROCGEN_AMD_CUTS=$cuts b $g; done; done
