# round 5, second GPU call: what changed since the first (per-env distribution_mode, per-chunk observation landing, device-side render order),
# most important first:
#   1 smoke, the new / changed GPU tests alone (fast feedback), bench line
#   2 host-landed rate: per-chunk observation copies off / on x launch chunks 2 / 4 / 8
#   3 render order on the device (if built): steady-state bench off / on, kernel trace + FETCH_SIZE on
#   4 the whole GPU suite with four workers, N more times (fatal log kept), then state I/O timing
# usage: bash tools/gpu/r5_call2.sh [tag] [suite runs]
TAG=${1:-r5c2}
RUNS=${2:-5}
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
export PROCGEN_AMD_FATAL_LOG=$R/gpurun_out/${TAG}_fatal.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/${TAG}_smoke.txt
timeout 900 python -m pytest tests/test_state_wire_format.py tests/test_gpu_sharing_stress.py tests/test_multi_gpu_paths.py tests/test_render_human.py -q -m gpu -n 4 2>&1 | tail -6 | tee gpurun_out/${TAG}_pytest_changed.log
timeout 600 python -m pytest tests/test_gpu_parity_at_scale.py -q -m gpu -n 4 -k "eight_device_shards or sixteen_game or separately" 2>&1 | tail -4 | tee -a gpurun_out/${TAG}_pytest_changed.log
python bench.py --no-cpu-baseline 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-1500 gpurun_out/${TAG}_bench.json
# ---- 2 host-landed observations (the unmodified ABI): one copy behind the step vs per-chunk copies, launch chunks 2 / 4 / 8
for oc in 0 1; do for ch in 2 4 8; do
  PROCGEN_AMD_OBS_CHUNK_COPY=$oc PROCGEN_AMD_CHUNKS=$ch python bench.py --host-landed --steps 30 --warmup 5 --steady-warmup 0 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('obs_chunk_copy=$oc chunks=$ch', round(d['value']/1e6,3), 'M steps/s host-landed,', d['ms_per_step'], 'ms/step')"
done; done 2>&1 | tee gpurun_out/${TAG}_host_landed_ab.txt
# same-box A/B against the round-4 library (is the 1 % the first call saw gone with the host-mapped error record?)
[ -f procgen_amd/csrc/build_r04/libenv.so ] && timeout 300 python tools/gpu/ab_bench.py procgen_amd/csrc/build_r04,procgen_amd/csrc/build coinrun,bigfish,maze 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_ab_vs_r04.txt
# ---- 3 render order, sorted on the device: same frames, device ms per step off / K=16 / K=64 after 1200 steps
timeout 900 python tools/gpu/render_order_ab.py coinrun,climber,ninja,jumper,maze,heist,caveflyer,miner,bigfish 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_render_order_ab.txt
# ---- 3b what the waves wait on (hardware counters)
bash tools/gpu/stall_counters.sh ${TAG}_stall 2>&1 | tail -45
cd $R
# ---- 4 the suite under GPU sharing
for i in $(seq 1 $RUNS); do
  timeout 900 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -8 > gpurun_out/${TAG}_pytest_parallel_$i.log
  tail -2 gpurun_out/${TAG}_pytest_parallel_$i.log
done
[ -f $PROCGEN_AMD_FATAL_LOG ] && { echo "--- fatal log"; grep -v "use_generated_assets\|another distribution_mode\|does not have" $PROCGEN_AMD_FATAL_LOG | cut -c1-3000; }
timeout 300 python tools/gpu/state_io_timing.py coinrun 65536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_state_io.txt
