R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r2_full_pytest.log
python bench.py --game all16 --num-envs 16384 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2_bench_all16_joint_16384.json; cut -c1-200 gpurun_out/r2_bench_all16_joint_16384.json
