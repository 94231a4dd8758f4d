# round 6, call 15: display list v6 (records hold 192 commands: frames with more than 64 sprites stay on the rasterizer; slow frames drawn by libenv_observe on a host-mapped flag): probe, tests, A/B incl. the release build, bench, 16 games
TAG=${1:-r6c15}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export PROCGEN_AMD_FATAL_LOG=$R/gpurun_out/${TAG}_fatal.log
python tools/gpu/dl_probe.py coinrun 65536 200 2>&1 | grep -v amdgpu | tail -6 | tee gpurun_out/${TAG}_dl_probe.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_state_wire_format.py -q -m gpu -x -n 4 -k "display_list or (coinrun and not protocol_at_its_own) or batched or launch_shape or option_surface or restored_envs_keep" 2>&1 | tail -4 | tee gpurun_out/${TAG}_pytest_dl.log
timeout 900 python tools/gpu/ab_bench.py procgen_amd/csrc/build_r05,procgen_amd/csrc/build,procgen_amd/csrc/build_rel coinrun,bigfish,bossfight 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_ab.txt
timeout 900 python tools/gpu/ab_bench.py procgen_amd/csrc/build_r05,procgen_amd/csrc/build maze,miner,climber,chaser 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/${TAG}_ab.txt
python bench.py --no-cpu-baseline 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-330 gpurun_out/${TAG}_bench.json
timeout 1200 python -m pytest tests -q -m gpu -n 4 -k "not protocol_at_its_own_length" 2>&1 | tail -8 > gpurun_out/${TAG}_pytest.log; tail -4 gpurun_out/${TAG}_pytest.log
