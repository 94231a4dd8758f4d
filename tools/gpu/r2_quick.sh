R=$GRAFT_REPO_ROOT
cd $R
python tools/gpu/ab_bench.py tools/gpu/ab/libenv_r01.so,procgen_amd/csrc/build/libenv.so leaper 2>&1 | grep -v amdgpu.ids
python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['host_landed'])"
python bench.py --game all16 --num-envs 16384 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('joint', j['value'], j['ms_per_step'])"
python -m pytest tests/test_gpu_parity.py -x -q -k "leaper or full_size or device_resident" 2>&1 | tail -2
