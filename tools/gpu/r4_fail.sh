R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest "tests/test_gpu_device_math.py::test_caveflyer_exhaust_puff_position_at_every_angle_where_the_doubles_differ" -q -s 2>&1 | tail -15 | tee gpurun_out/r4_fail_puff.log
timeout 600 python -m pytest tests/test_render_human.py -m gpu -q -k "gen" 2>&1 | tail -15 | tee gpurun_out/r4_fail_gen.log
timeout 600 python -m pytest tests/test_multi_gpu_paths.py -m gpu -q -s 2>&1 | tail -30 | tee gpurun_out/r4_fail_multi.log
dmesg 2>/dev/null | tail -5
