# round 6, call 19: raster ablation on the current build (where did +850 VALU per frame come from since call 11?)
TAG=${1:-r6c19}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for f in ${FLAGS:-0 2 4}; do
  PROCGEN_AMD_DEBUG=$f timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --kernel-trace -d /tmp/${TAG}_f$f -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-host-landed --no-traffic > $R/gpurun_out/${TAG}_f$f.log 2>&1
  python $R/tests/tools/rocpd_summary.py $(find /tmp/${TAG}_f$f -name "*.db" | head -1) > $R/gpurun_out/${TAG}_f$f.csv 2>&1
  rm -rf /tmp/${TAG}_f$f
  python - $R/gpurun_out/${TAG}_f$f.csv $f <<'PY'
import sys
rows=[l.strip().split(',') for l in open(sys.argv[1])]
d={};dur=None
for r in rows:
    if '6rasterI' in r[0]:
        if r[1].startswith('SQ_'): d[r[1]]=float(r[3])
        elif dur is None and len(r)>3: dur=r[2]
w=d['SQ_WAVES']
print("debug=%-3s raster avg_us=%-8s VALU/wave=%6.0f SALU/wave=%6.0f LDS/wave=%5.0f VMEM/wave=%5.0f cycles/wave=%7.0f"%(sys.argv[2],dur,d['SQ_INSTS_VALU']/w,d['SQ_INSTS_SALU']/w,d['SQ_INSTS_LDS']/w,d['SQ_INSTS_VMEM_RD']/w,4*d['SQ_WAVE_CYCLES']/w))
PY
done | tee $R/gpurun_out/${TAG}_ablation.txt
