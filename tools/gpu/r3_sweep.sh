# round 3, after the step kernel got 2.7x faster: is the two-chunk 75 / 25 cut still the right launch shape?
R=$GRAFT_REPO_ROOT
cd $R
run() { python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']/1e6,2), d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"; }
echo "default: $(run)"
for c in 1 3; do echo "chunks $c: $(PROCGEN_AMD_CHUNKS=$c run)"; done
for p in 40 50 60 67 85 90; do echo "first_pct $p: $(PROCGEN_AMD_FIRST_PCT=$p run)"; done
for o in 1 2 3; do echo "order $o: $(PROCGEN_AMD_ORDER=$o run)"; done
echo "no render (debug 16): $(PROCGEN_AMD_DEBUG=16 run)"
echo "no tier-0 step (debug 32): $(PROCGEN_AMD_DEBUG=32 run)"
for g in bigfish starpilot maze; do for p in 50 75 90; do echo "$g first_pct $p: $(PROCGEN_AMD_FIRST_PCT=$p python bench.py --game $g --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']/1e6,2))")"; done; done
