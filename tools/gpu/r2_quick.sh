R=$GRAFT_REPO_ROOT
cd $R
python tools/gpu/ab_bench.py tools/gpu/ab/libenv_r01.so,procgen_amd/csrc/build/libenv.so coinrun,bigfish 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gpu_parity.py -x -q -k "coinrun and (golden or parity)" 2>&1 | tail -2
