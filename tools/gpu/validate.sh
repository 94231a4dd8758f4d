# round-end check on one MI355X: the whole GPU suite, the default bench line, the one-GPU share of configs[4]
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py 2>/dev/null | tail -1 > gpurun_out/final_bench_coinrun.json; python -c "import json; d=json.load(open('gpurun_out/final_bench_coinrun.json')); print(d['value'], d['ms_per_step'], d['roofline'], d['cpu_baseline'])"
python bench.py --game all16 --num-envs 16384 --steps 120 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final_bench_all16_16384.json; python -c "import json; d=json.load(open('gpurun_out/final_bench_all16_16384.json')); print('all16 16384', d['value'], d['ms_per_step'])"
python tools/gpu/host_timing.py 2>&1 | grep -v amdgpu.ids
