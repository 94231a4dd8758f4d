R=$GRAFT_REPO_ROOT
cd $R
python tools/gpu/ab_bench.py tools/gpu/ab/libenv_r01.so,procgen_amd/csrc/build/libenv.so coinrun 2>&1 | grep -v amdgpu.ids
python -m pytest tests -m gpu -x -q -k "coinrun or entity_table or arena_tiers or full_size or sixteen or joint_games" 2>&1 | tail -2
