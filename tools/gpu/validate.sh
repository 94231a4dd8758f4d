# round-end check on one MI355X: smoke(), the whole GPU suite, the default bench line, the one-GPU share of configs[4], all 16 games
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py 2>/dev/null | tail -1 > gpurun_out/final_bench_coinrun.json; python -c "import json; d=json.load(open('gpurun_out/final_bench_coinrun.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])"
python bench.py --game all16 --num-envs 16384 --steps 120 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final_bench_all16_16384.json; python -c "import json; d=json.load(open('gpurun_out/final_bench_all16_16384.json')); print('all16 16384', d['value'], d['ms_per_step'])"
bash tools/gpu/bench16.sh 2>&1 | tee gpurun_out/final_bench16.log
