# round 6: per-channel L2 request counts (unreduced TCC_REQ / TCC_BUSY) of raster<BigFish> with the launch order off / on: is the order's cost a hot channel?
TAG=${1:-r6tccch}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for k in 0 16; do
  PROCGEN_AMD_RENDER_ORDER=$k timeout 200 rocprofv3 --pmc TCC_REQ TCC_BUSY --kernel-trace -d /tmp/${TAG}_$k -o p -- python $R/bench.py --game bigfish --steps 8 --warmup 2 --steady-warmup 300 --no-cpu-baseline --no-host-landed --no-traffic > $R/gpurun_out/${TAG}_$k.log 2>&1
  DB=$(find /tmp/${TAG}_$k -name "*.db" | head -1)
  python - $DB $k <<'PY' | tee -a $R/gpurun_out/${TAG}_table.txt
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
T = {r[0].rsplit("_0000", 1)[0]: r[0] for r in c.execute("select name from sqlite_master where type='table'")}
cols = [r[1] for r in c.execute(f"pragma table_info({T['rocpd_info_pmc']})")]
print("order", sys.argv[2], "info_pmc columns:", cols)
q = (f"select p.name, p.id, sum(e.value) from {T['rocpd_pmc_event']} e join {T['rocpd_info_pmc']} p on e.pmc_id=p.id join {T['rocpd_kernel_dispatch']} d on e.event_id=d.event_id "
     f"join {T['rocpd_info_kernel_symbol']} s on d.kernel_id=s.id where s.kernel_name like '%raster%' group by p.name, p.id")
rows = list(c.execute(q))
by = {}
for name, pid, v in rows:
    by.setdefault(name, []).append(v)
for name, vs in by.items():
    vs = sorted(vs)
    print(f"order {sys.argv[2]} {name}: {len(vs)} instances, sum {sum(vs):.3e}, min {vs[0]:.3e}, median {vs[len(vs)//2]:.3e}, max {vs[-1]:.3e}, max/mean {vs[-1] / (sum(vs) / len(vs)):.2f}")
PY
  rm -rf /tmp/${TAG}_$k
done
