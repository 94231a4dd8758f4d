R=$GRAFT_REPO_ROOT
cd $R
python tools/gpu/ab_bench.py tools/gpu/ab/libenv_r01.so,procgen_amd/csrc/build/libenv.so coinrun,bigfish,starpilot,maze,leaper,jumper 2>&1 | grep -v amdgpu.ids
