# round 3: level generator stage cycles (PROCGEN_AMD_DEBUG=2064: phase counters + reset marks) for jumper / caveflyer, then the joint share
R=$GRAFT_REPO_ROOT
cd $R
for g in jumper caveflyer; do
  PROCGEN_AMD_DEBUG=2064 python bench.py --game $g --steps 40 --warmup 10 --no-cpu-baseline 2>&1 | grep -i "reset marks\|mark " | head -14
done
for g in leaper jumper caveflyer maze heist; do python bench.py --game $g --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$g', round(d['value']/1e6,2), 'M steps/s')"; done
python bench.py --game all16 --num-envs 16384 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3_bench_all16_joint_16384.json; python -c "import json; d=json.load(open('gpurun_out/r3_bench_all16_joint_16384.json')); print('all16 joint', round(d['value']/1e6,2), 'M steps/s', d['ms_per_step'], 'ms')"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt -o kt -- python $R/bench.py --game all16 --num-envs 16384 --steps 40 --warmup 10 --no-cpu-baseline > $R/gpurun_out/r3_joint_kt.log 2>&1
python $R/tests/tools/rocpd_summary.py $(find $R/gpurun_out/kt -name "*.db" | head -1) > $R/gpurun_out/r3_joint_kernel_trace.csv 2>&1
python $R/tests/tools/rocpd_timeline.py $(find $R/gpurun_out/kt -name "*.db" | head -1) w2.5 > $R/gpurun_out/r3_joint_timeline.txt 2>&1
rm -rf $R/gpurun_out/kt
sort -t, -k3 -n -r $R/gpurun_out/r3_joint_kernel_trace.csv | cut -c1-120 | head -14
