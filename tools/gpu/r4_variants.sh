# round 4: same-box A/B of renderer variants (tools/gpu/ab/libenv_*.so, not tracked), coinrun 65536 envs, + their phase tables
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
LIBS=$(ls tools/gpu/ab/libenv_*.so | tr '\n' ',' | sed 's/,$//')
python tools/gpu/ab_bench.py $LIBS coinrun 2>&1 | tee gpurun_out/r4_variants.txt
for l in $(ls tools/gpu/ab/libenv_*.so); do
  echo "== $l" | tee -a gpurun_out/r4_variants_phase.txt
  PROCGEN_AMD_DEBUG=2048 python -c "
import sys; sys.argv=['x','$l','coinrun']
sys.path.insert(0,'tools/gpu')
import ab_bench
ab_bench.run('$l','coinrun',65536,steps=60,warm=10)
" 2>&1 | grep -A12 "render kernel" | tee -a gpurun_out/r4_variants_phase.txt
done
