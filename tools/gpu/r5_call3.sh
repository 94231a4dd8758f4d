# round 5, third GPU call (short): two more occupancy candidates as a variant build (tools: a worktree with the two policy lines changed, the two
# objects compiled by hand and linked with the default build's other objects into procgen_amd/csrc/build_try): bossfight with 32 rotation
# records instead of 64 (arena 11984 -> 8912 B, frames with more turned bullets take the per-band path), climber's renderer at five waves
# per SIMD (102 -> 96 VGPRs, 12 B of scratch).  Parity subset on the variant first, then the same-box A/B.
TAG=${1:-r5c3}
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
PROCGEN_AMD_LIB_DIR=$R/procgen_amd/csrc/build_try timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 4 -k "bossfight or climber or oracle" 2>&1 | tail -3 | tee gpurun_out/${TAG}_parity_build_try.log
timeout 600 python tools/gpu/ab_bench.py procgen_amd/csrc/build,procgen_amd/csrc/build_try bossfight,climber 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_try_ab.txt
# the tests added or changed since the second call, then the state I/O timing with the parallel serializer, one more four-worker suite run
timeout 900 python -m pytest tests/test_state_wire_format.py tests/test_gpu_parity.py tests/test_gpu_parity_at_scale.py -q -m gpu -n 4 -k "refused or other_indices or launch_order or chunk_by_chunk or lds_dma" 2>&1 | tail -4 | tee gpurun_out/${TAG}_pytest_changed.log
timeout 300 python tools/gpu/state_io_timing.py coinrun 65536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_state_io.txt
python bench.py --no-cpu-baseline 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-400 gpurun_out/${TAG}_bench.json; python -c "import json; d=json.load(open('gpurun_out/${TAG}_bench.json')); print({k: d[k] for k in ('host_landed', 'cold_start', 'steady_state')}); print(d['roofline'])"
export PROCGEN_AMD_FATAL_LOG=$R/gpurun_out/${TAG}_fatal.log
timeout 900 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -4 | tee gpurun_out/${TAG}_pytest_parallel.log
