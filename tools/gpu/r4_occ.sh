# round 4: does a fifth wave per SIMD pay?  coinrun's render kernel at 96 VGPRs / 8068 B LDS (tools/gpu/ab/libenv_occ5.so) against the 107-VGPR build
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
python tools/gpu/ab_bench.py tools/gpu/ab/libenv_r03.so,procgen_amd/csrc/build/libenv.so,tools/gpu/ab/libenv_occ5.so coinrun 2>&1 | tee gpurun_out/r4_occ.txt
python tools/gpu/ab_bench.py tools/gpu/ab/libenv_r03.so,procgen_amd/csrc/build/libenv.so,tools/gpu/ab/libenv_occ5.so coinrun 2>&1 | tee -a gpurun_out/r4_occ.txt
