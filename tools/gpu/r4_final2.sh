# round 4, last checkpoint (after the early download of the small outputs): parity at scale, the default bench line, the 16 games, the joint share
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
TAG=r4_final2
timeout 900 python -m pytest tests/test_gpu_parity_at_scale.py tests/test_torch_view.py tests/test_multi_gpu_paths.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/${TAG}_pytest.log
python bench.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-200 gpurun_out/${TAG}_bench.json
for g in coinrun bigfish maze climber miner starpilot fruitbot leaper plunder heist ninja dodgeball bossfight chaser caveflyer jumper; do python bench.py --game $g --steps 120 --warmup 20 --no-cpu-baseline --steady-warmup 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['config']['workload'].split()[0], round(d['value']/1e6,2))"; done 2>&1 | tee gpurun_out/${TAG}_bench16.log
python bench.py --game all16 --num-envs 16384 --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_all16_joint_16384.json; cut -c1-200 gpurun_out/${TAG}_bench_all16_joint_16384.json
python bench.py --game bigfish --steps 150 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_bigfish_65536.json
python bench.py --game starpilot --num-envs 32768 --steps 150 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_starpilot_32768.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_kt -o kt -- python $R/bench.py --steps 40 --warmup 1500 --no-cpu-baseline --steady-warmup 0 > $R/gpurun_out/${TAG}_kt.log 2>&1
DB=$(find $R/gpurun_out/${TAG}_kt -name "*.db" | head -1)
python $R/tests/tools/rocpd_summary.py $DB > $R/gpurun_out/${TAG}_kernel_trace.csv 2>&1
python $R/tests/tools/rocpd_timeline.py $DB 3 > $R/gpurun_out/${TAG}_timeline.txt 2>&1
rm -rf $R/gpurun_out/${TAG}_kt
head -8 $R/gpurun_out/${TAG}_kernel_trace.csv; tail -3 $R/gpurun_out/${TAG}_timeline.txt
