R=$GRAFT_REPO_ROOT
cd $R
b() { python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step'])"; }
python -m pytest tests/test_gpu_parity.py -x -q -k "(coinrun and (golden or parity or forced or state)) or entity_table or arena_tiers" 2>&1 | tail -2
echo "default (16/2, 2 chunks)"; b
echo "no lane"; PROCGEN_AMD_DEBUG=4096 b
echo "8/1"; PROCGEN_AMD_LANE_ENTS=8 PROCGEN_AMD_LANE_SMART=1 b
echo "12/1"; PROCGEN_AMD_LANE_ENTS=12 PROCGEN_AMD_LANE_SMART=1 b
echo "16/1"; PROCGEN_AMD_LANE_ENTS=16 PROCGEN_AMD_LANE_SMART=1 b
echo "12/2"; PROCGEN_AMD_LANE_ENTS=12 PROCGEN_AMD_LANE_SMART=2 b
echo "16/3"; PROCGEN_AMD_LANE_ENTS=16 PROCGEN_AMD_LANE_SMART=3 b
echo "24/2"; PROCGEN_AMD_LANE_ENTS=24 PROCGEN_AMD_LANE_SMART=2 b
echo "1 chunk"; PROCGEN_AMD_CHUNKS=1 b
echo "3 chunks"; PROCGEN_AMD_CHUNKS=3 b
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt -o kt -- python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $R/gpurun_out/r2_lane_kt.log 2>&1
python $R/tests/tools/rocpd_summary.py $(find $R/gpurun_out/kt -name "*.db" | head -1) > $R/gpurun_out/r2_lane_kernel_trace.csv 2>&1
rm -rf $R/gpurun_out/kt
head -10 $R/gpurun_out/r2_lane_kernel_trace.csv
