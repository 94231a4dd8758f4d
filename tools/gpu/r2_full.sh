R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r2_full_pytest.log
python tools/gpu/ab_bench.py tools/gpu/ab/libenv_r01.so,procgen_amd/csrc/build/libenv.so coinrun,bigfish,jumper 2>&1 | grep -v amdgpu.ids
python bench.py --steps 100 --warmup 20 2>/dev/null | tail -1 > gpurun_out/r2_bench_default.json; cut -c1-220 gpurun_out/r2_bench_default.json
PROCGEN_AMD_FAKE_DEVICES=1 python bench.py --steps 50 --warmup 10 --num-envs 32768 --devices-in-process 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
