# round 4, closing checkpoint (after the LDS diet): A/B of the five-wave games, parity, the default bench line, the 16 games, the config shares
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
TAG=r4_final3
python tools/gpu/ab_bench.py tools/gpu/ab/libenv_r03.so,procgen_amd/csrc/build/libenv.so maze,miner,ninja 2>&1 | awk 'NR%2==0' | tee gpurun_out/${TAG}_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 4 2>&1 | tail -3 | tee gpurun_out/${TAG}_pytest.log
python bench.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-200 gpurun_out/${TAG}_bench.json
for g in coinrun bigfish maze climber miner starpilot fruitbot leaper plunder heist ninja dodgeball bossfight chaser caveflyer jumper; do python bench.py --game $g --steps 120 --warmup 20 --no-cpu-baseline --steady-warmup 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['config']['workload'].split()[0], round(d['value']/1e6,2))"; done 2>&1 | tee gpurun_out/${TAG}_bench16.log
python bench.py --game all16 --num-envs 16384 --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_all16_joint_16384.json; cut -c1-200 gpurun_out/${TAG}_bench_all16_joint_16384.json
python bench.py --game bigfish --steps 150 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_bigfish_65536.json
python bench.py --game starpilot --num-envs 32768 --steps 150 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_starpilot_32768.json
