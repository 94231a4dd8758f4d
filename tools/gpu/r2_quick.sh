R=$GRAFT_REPO_ROOT
cd $R
python tools/gpu/ab_bench.py tools/gpu/ab/libenv_r01.so,procgen_amd/csrc/build/libenv.so leaper 2>&1 | grep -v amdgpu.ids | awk 'NR%2==0'
python bench.py --game all16 --num-envs 16384 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('joint', j['value'], j['ms_per_step'])"
python -m pytest tests -m gpu -x -q -k "leaper" 2>&1 | tail -2
