R=$GRAFT_REPO_ROOT
cd $R
python tools/gpu/ab_bench.py tools/gpu/ab/libenv_r01.so,procgen_amd/csrc/build/libenv.so jumper,caveflyer,leaper 2>&1 | grep -v amdgpu.ids | awk 'NR%2==0'
PROCGEN_AMD_FIRST_PCT=50 python tools/gpu/ab_bench.py procgen_amd/csrc/build/libenv.so jumper,caveflyer,leaper 2>&1 | grep -v amdgpu.ids | awk 'NR%2==0'
python -m pytest tests/test_gpu_parity.py -x -q -k "(jumper or caveflyer or leaper or coinrun) and (parity_with or golden)" 2>&1 | tail -2
