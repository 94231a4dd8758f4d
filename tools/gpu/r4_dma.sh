# round 4: the background by LDS-DMA, merged seam stages, merged z passes: parity on the device (all 16 games draw through these paths), then same-box A/B
# against the round-3 library and the switches (PROCGEN_AMD_DEBUG 131072: no DMA, 262144: no seam merge)
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_render_human.py -m gpu -x -q -n 4 2>&1 | tail -4 | tee gpurun_out/r4_dma_pytest.log
python tools/gpu/ab_bench.py tools/gpu/ab/libenv_r03.so,procgen_amd/csrc/build/libenv.so coinrun 2>&1 | tee gpurun_out/r4_dma_ab.txt
for f in 131072 262144 393216; do echo "PROCGEN_AMD_DEBUG=$f"; PROCGEN_AMD_DEBUG=$f python tools/gpu/ab_bench.py procgen_amd/csrc/build/libenv.so coinrun 2>&1; done | tee -a gpurun_out/r4_dma_ab.txt
python tools/gpu/ab_bench.py tools/gpu/ab/libenv_r03.so,procgen_amd/csrc/build/libenv.so bigfish,starpilot,maze,fruitbot,leaper,dodgeball,bossfight,heist,jumper,miner,ninja 2>&1 | tee -a gpurun_out/r4_dma_ab.txt
PROCGEN_AMD_DEBUG=2048 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | grep -A28 "phase cycles" | tee gpurun_out/r4_dma_phase.txt
