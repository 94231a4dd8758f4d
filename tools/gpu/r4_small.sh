# round 4 (kept as the record of an experiment that did not pay: zero-copy host arrays for small handles measured the same 0.187 ms per
# 64-env step as the copy-based step and made the test suite 2.5x slower; removed): zero-copy steps of small handles: whole GPU suite (every small test handle now runs in that mode), the small-handle table with and
# without it, the 16-game joint share (its 16 parts are small handles)
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
export PROCGEN_AMD_FATAL_LOG=$R/gpurun_out/r4_small_fatal.log
timeout 1500 python -m pytest tests -m gpu -q -n 4 --tb=short 2>&1 | tail -15 | tee gpurun_out/r4_small_pytest.log
echo "== zero-copy" | tee gpurun_out/r4_small_handles.txt; python tools/gpu/small_handles.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4_small_handles.txt
echo "== copies (PROCGEN_AMD_NO_ZEROCOPY=1)" | tee -a gpurun_out/r4_small_handles.txt; PROCGEN_AMD_NO_ZEROCOPY=1 python tools/gpu/small_handles.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4_small_handles.txt
python bench.py --game all16 --num-envs 16384 --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | tee gpurun_out/r4_small_joint.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('all16 joint', round(d['value']/1e6,2), 'M steps/s', d['ms_per_step'], 'ms')"
PROCGEN_AMD_NO_ZEROCOPY=1 python bench.py --game all16 --num-envs 16384 --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('all16 joint (copies)', round(d['value']/1e6,2), 'M steps/s', d['ms_per_step'], 'ms')"
cat gpurun_out/r4_small_fatal.log 2>/dev/null | grep -v "use_generated_assets\|distribution_mode" | tail
