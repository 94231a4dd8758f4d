R=$GRAFT_REPO_ROOT
cd $R
python bench.py --game all16 --num-envs 16384 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('joint', j['value'], j['ms_per_step'])"
python tools/gpu/ab_bench.py procgen_amd/csrc/build/libenv.so jumper,leaper 2>&1 | grep -v amdgpu.ids | awk 'NR%2==0'
python -m pytest tests/test_gpu_parity.py -x -q -k "(jumper or leaper) and golden" 2>&1 | tail -1
