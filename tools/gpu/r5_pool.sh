# round 5: the rotation-record pool build variant (pg_render.h PG_ROT_POOL, prepared and emulation-checked at the end of round 4, never
# run on a device) against the default build, same box, alternating.  Build the variants in the container BEFORE the gpurun call
# (3.5 min each on 8 cores; the directories travel with the snapshot, they are git-ignored):
#   make -s -j8 -C procgen_amd/csrc ARCH=gfx950 BUILD=build_pool16   EXTRA="-DPG_ROT_POOL=16"
#   make -s -j8 -C procgen_amd/csrc ARCH=gfx950 BUILD=build_pool16w4 EXTRA="-DPG_ROT_POOL=16 -DPG_RENDER_WAVES=4"
# (the w4 build also moves jumper: 132 -> 128 VGPRs with 12 B of scratch, 12 -> 16 frames per CU.)
# First the parity subset on each variant (PROCGEN_AMD_LIB_DIR), then the A/B of the games with rotation records.
# usage: bash tools/gpu/r5_pool.sh [tag]
TAG=${1:-r5_pool}
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
LIBS=procgen_amd/csrc/build
for v in build_pool16 build_pool16w4; do
  [ -f procgen_amd/csrc/$v/libenv.so ] || { echo "missing procgen_amd/csrc/$v/libenv.so"; continue; }
  LIBS=$LIBS,procgen_amd/csrc/$v
  PROCGEN_AMD_LIB_DIR=$R/procgen_amd/csrc/$v timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "oracle or fixture" 2>&1 | tail -3 | tee gpurun_out/${TAG}_parity_$v.log
done
timeout 1500 python tools/gpu/ab_bench.py $LIBS heist,caveflyer,plunder,starpilot,dodgeball,leaper,fruitbot,bossfight,jumper 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_ab.txt
