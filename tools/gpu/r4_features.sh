# round 4: new GPU tests (torch views, both multi-GPU launch paths on the one device), then the default bench line with steady_state
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_torch_view.py tests/test_multi_gpu_paths.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r4_features_pytest.log
python bench.py --no-cpu-baseline 2>gpurun_out/r4_features_bench.err | tail -1 | tee gpurun_out/r4_features_bench.json
tail -5 gpurun_out/r4_features_bench.err
