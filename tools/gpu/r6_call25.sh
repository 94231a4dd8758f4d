# round 6, call 25: raster with the pull form's column / row / type tables in five vector registers and the band buffer over their LDS words
# (coinrun 5968 -> 4688 bytes of LDS = four granules of 1280: 25 -> 32 frames per CU; bigfish 5472 -> 4432): tests, same-box A/B against the
# build before (build_prev) and round 5, bench line, raster counters
TAG=${1:-r6c25}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -n 4 -k "display_list or golden_rollout or parity_with_oracle or batched" 2>&1 | tail -3 | tee gpurun_out/${TAG}_pytest.log
timeout 900 python tools/gpu/ab_bench.py procgen_amd/csrc/build_r05,procgen_amd/csrc/build_prev,procgen_amd/csrc/build coinrun,bigfish,maze,climber 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_ab.txt
python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-330 gpurun_out/${TAG}_bench.json
FLAGS=0 bash tools/gpu/r6_call19.sh ${TAG}
