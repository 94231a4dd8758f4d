cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in 1 2; do
  if [ $pass = 1 ]; then C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES"; else C="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES"; fi
  rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmc$pass -o p -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline > $R/gpurun_out/pmc$pass.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for p in (1,2):
    fs = glob.glob(f"gpurun_out/pmc{p}/**/*counter_collection.csv", recursive=True)
    print("pass", p, fs)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVES","SQ_WAIT_ANY"): cnt[k] += 1
    for k in agg:
        print(k, cnt[k], {a: round(b/max(cnt[k],1)) for a,b in agg[k].items()})
PY
rm -rf gpurun_out/pmc1 gpurun_out/pmc2
