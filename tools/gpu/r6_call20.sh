# round 6, call 20: raster with its record header in a lane register (local v_readlane instead of spilled scalar tuples): tests, A/B, ablation
TAG=${1:-r6c20}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -n 4 -k "display_list or golden_rollout or parity_with_oracle or batched" 2>&1 | tail -3 | tee gpurun_out/${TAG}_pytest.log
timeout 900 python tools/gpu/ab_bench.py procgen_amd/csrc/build_r05,procgen_amd/csrc/build coinrun,bigfish,maze,miner,climber,chaser 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_ab.txt
python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-330 gpurun_out/${TAG}_bench.json
bash tools/gpu/r6_call19.sh ${TAG}
