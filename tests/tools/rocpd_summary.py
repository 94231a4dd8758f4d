"""Prints per-kernel duration stats (and PMC averages, if any) from a rocprofv3 rocpd sqlite database."""
import sqlite3
import sys


def tabs(c):
    return {r[0].rsplit("_0000", 1)[0]: r[0] for r in c.execute("select name from sqlite_master where type='table'")}


def main(path):
    c = sqlite3.connect(path)
    T = tabs(c)
    kd, ks = T["rocpd_kernel_dispatch"], T["rocpd_info_kernel_symbol"]
    print(f"# {path}")
    print("kernel,calls,avg_us,min_us,max_us,total_ms,lds_bytes,vgpr,sgpr")
    q = (f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3, sum(d.end-d.start)/1e6, "
         f"max(d.group_segment_size), max(s.arch_vgpr_count), max(s.sgpr_count) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 6 desc")
    for r in c.execute(q):
        print(",".join(str(round(x, 3)) if isinstance(x, float) else str(x) for x in r))
    pe, pi = T.get("rocpd_pmc_event"), T.get("rocpd_info_pmc")
    if pe and pi:
        q = (f"select s.kernel_name, p.name, count(*), sum(e.value) from {pe} e join {pi} p on e.pmc_id=p.id join {kd} d on e.event_id=d.event_id "
             f"join {ks} s on d.kernel_id=s.id group by s.kernel_name, p.name")
        rows = list(c.execute(q))
        if rows:
            print("kernel,counter,samples,sum")
            for r in rows:
                print(",".join(str(x) for x in r))


if __name__ == "__main__":
    for p in sys.argv[1:]:
        main(p)
