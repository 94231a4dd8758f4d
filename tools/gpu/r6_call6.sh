# round 6, call 6: the one-kernel renderer with the cell-row-major grid pass (PROCGEN_AMD_DISPLAY_LIST=0) against the display list and round 5; launch shape re-sweep
TAG=${1:-r6c6}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
echo "== display list off (one-kernel renderer + cellrows_pass)" | tee gpurun_out/${TAG}_ab.txt
PROCGEN_AMD_DISPLAY_LIST=0 timeout 1200 python tools/gpu/ab_bench.py procgen_amd/csrc/build_r05,procgen_amd/csrc/build coinrun,climber,chaser,ninja,jumper,heist,maze,miner 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/${TAG}_ab.txt
echo "== display list on" | tee -a gpurun_out/${TAG}_ab.txt
timeout 900 python tools/gpu/ab_bench.py procgen_amd/csrc/build_r05,procgen_amd/csrc/build coinrun,climber,chaser,maze,miner,bigfish 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/${TAG}_ab.txt
echo "== launch shape (display list on, coinrun): first chunk share" | tee -a gpurun_out/${TAG}_ab.txt
for p in 50 60 75 85; do echo "FIRST_PCT=$p"; PROCGEN_AMD_FIRST_PCT=$p timeout 300 python tools/gpu/ab_bench.py procgen_amd/csrc/build coinrun 2>&1 | grep -v amdgpu.ids; done | tee -a gpurun_out/${TAG}_ab.txt
for c in 3 4; do echo "CHUNKS=$c"; PROCGEN_AMD_CHUNKS=$c timeout 300 python tools/gpu/ab_bench.py procgen_amd/csrc/build coinrun 2>&1 | grep -v amdgpu.ids; done | tee -a gpurun_out/${TAG}_ab.txt
python bench.py --no-cpu-baseline 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-300 gpurun_out/${TAG}_bench.json; python -c "
import json; j=json.load(open('gpurun_out/${TAG}_bench.json')); print(j['roofline'])"
PROCGEN_AMD_DISPLAY_LIST=0 python bench.py --no-cpu-baseline --no-traffic --no-host-landed 2>/dev/null | tail -1 | cut -c1-300
