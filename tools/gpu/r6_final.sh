# round 6, closing checkpoint, most important first: the whole GPU suite SERIALLY (as the driver runs it, timed), the default bench line with
# cpu_baseline and the live traffic passes, steady-state kernel trace + per-dispatch timeline, SQ instruction counters of the frame and step
# kernels, the 16 games alone, the joint handle's one-GPU share, config shares, state I/O, render order on / off for bigfish.
# usage: bash tools/gpu/r6_final.sh [tag]
TAG=${1:-r6_final}
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
export PROCGEN_AMD_FATAL_LOG=$R/gpurun_out/${TAG}_fatal.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/${TAG}_smoke.txt
( time timeout 2400 python -m pytest tests -x -q -m gpu ) 2>&1 | tail -8 | tee gpurun_out/${TAG}_pytest.log
python bench.py 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-600 gpurun_out/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_kt -o kt -- python $R/bench.py --steps 64 --warmup 5 --no-cpu-baseline --no-host-landed --no-traffic > $R/gpurun_out/${TAG}_kt.log 2>&1
DB=$(find /tmp/${TAG}_kt -name "*.db" | head -1)
python $R/tests/tools/rocpd_summary.py $DB > $R/gpurun_out/${TAG}_kernel_trace.csv 2>&1
python $R/tests/tools/rocpd_timeline.py $DB > $R/gpurun_out/${TAG}_timeline.txt 2>&1
rm -rf /tmp/${TAG}_kt; head -9 $R/gpurun_out/${TAG}_kernel_trace.csv | cut -c1-170
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --kernel-trace -d /tmp/${TAG}_f0 -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-host-landed --no-traffic > $R/gpurun_out/${TAG}_f0.log 2>&1
python $R/tests/tools/rocpd_summary.py $(find /tmp/${TAG}_f0 -name "*.db" | head -1) > $R/gpurun_out/${TAG}_sq_insts.csv 2>&1
rm -rf /tmp/${TAG}_f0
cd $R
for g in coinrun bigfish maze climber miner starpilot fruitbot leaper plunder heist ninja dodgeball bossfight chaser caveflyer jumper; do python bench.py --game $g --steps 120 --warmup 20 --no-cpu-baseline --no-host-landed --no-traffic --steady-warmup 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['config']['workload'].split()[0], round(d['value']/1e6,2))"; done 2>&1 | tee gpurun_out/${TAG}_bench16.log
python bench.py --game all16 --num-envs 16384 --steps 100 --warmup 20 --no-cpu-baseline --steady-warmup 0 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_all16_joint_16384.json; cut -c1-220 gpurun_out/${TAG}_bench_all16_joint_16384.json
python bench.py --game bigfish --steps 150 --warmup 20 --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_bigfish_65536.json; cut -c1-200 gpurun_out/${TAG}_bench_bigfish_65536.json
python bench.py --game starpilot --num-envs 32768 --steps 150 --warmup 20 --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_starpilot_32768.json; cut -c1-200 gpurun_out/${TAG}_bench_starpilot_32768.json
timeout 600 python tools/gpu/state_io_timing.py 2>&1 | tail -4 | tee gpurun_out/${TAG}_state_io.txt
timeout 600 python tools/gpu/render_order_ab.py bigfish,coinrun 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_render_order_ab.txt
timeout 900 python -m pytest tests -q -m gpu -n 4 -k "not protocol_at_its_own_length" 2>&1 | tail -4 | tee gpurun_out/${TAG}_pytest_parallel.log
grep "fatal:" $PROCGEN_AMD_FATAL_LOG | cut -c1-150 | sed 's/\[pid [0-9]*\] //' | sort | uniq -c
bash tools/gpu/r6_tcc_channels.sh ${TAG}_tccch 2>&1 | tail -8
