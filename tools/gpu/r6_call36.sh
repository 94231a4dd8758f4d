# round 6, call 36: fruitbot and dodgeball declared gridless (DRAWS_GRID = false: their grids hold only SPACE -- no grid tables in the arena, no grid pass in
# the frame kernel), bossfight's rotation pool at 48 records: parity tests of the three games, same-box A/B against the build before (build_prev)
TAG=${1:-r6c36}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -x -n 4 -k "fruitbot or dodgeball or bossfight" 2>&1 | tail -3 | tee gpurun_out/${TAG}_pytest.log
for i in 1 2; do timeout 600 python tools/gpu/ab_bench.py procgen_amd/csrc/build_prev,procgen_amd/csrc/build fruitbot,dodgeball,bossfight 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/${TAG}_ab.txt
