R=$GRAFT_REPO_ROOT
cd $R
python tools/gpu/ab_bench.py tools/gpu/ab/libenv_r01.so,procgen_amd/csrc/build/libenv.so climber,ninja,dodgeball,chaser,caveflyer 2>&1 | grep -v amdgpu.ids | awk 'NR%2==0'
python -m pytest tests/test_gpu_parity.py -x -q -k "(climber or ninja or dodgeball or chaser or caveflyer) and (golden or parity_with or state_protocol)" 2>&1 | tail -2
