# round 6, call 28: the maze generator's LDS scratch sized per game (chaser 8328 -> 3056 bytes, heist 5488, maze 7952): step_tier0 arenas
# chaser 11 600 -> 6320, heist 10 736 -> 7888, maze 10 496 -> 10 112 bytes, i.e. 12 / 14 / 14 -> 16 workgroups per CU (the VGPR bound).
# Tests of the three games (+ jumper, whose generator embeds the maze scratch), same-box A/B against the build before (build_prev)
TAG=${1:-r6c28}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -n 4 -k "chaser or heist or maze or jumper" 2>&1 | tail -3 | tee gpurun_out/${TAG}_pytest.log
timeout 900 python tools/gpu/ab_bench.py procgen_amd/csrc/build_prev,procgen_amd/csrc/build chaser,heist,maze,jumper 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_ab.txt
