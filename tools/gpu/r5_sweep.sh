# round 5, last short call: launch-shape knobs in the STEADY state with the launch order on (rounds 2-3 swept them in the cold start):
# share of the first chunk, period of the launch-order rebuild.  Env-var level only (no build).
TAG=${1:-r5_sweep}
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-host-landed --steps 150 --warmup 20"
one() { $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', round(d['value']/1e6,2), 'M', d['ms_per_step'], 'ms/step, kernels', d['roofline']['kernel_ms_per_step'])"; }
one "default (first_pct 75, order 16)"
for p in 65 70 80 85; do PROCGEN_AMD_FIRST_PCT=$p one "first_pct=$p"; done
for k in 4 64; do PROCGEN_AMD_RENDER_ORDER=$k one "render_order=$k"; done
one "default again"
