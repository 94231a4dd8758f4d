"""Per-dispatch timeline from a rocprofv3 rocpd database: start / end (us, relative to the first dispatch shown), duration, queue and
kernel of the last N steps' dispatches -- names the chain of kernels a step's wall time hangs on.

    python tests/tools/rocpd_timeline.py <db> [steps]
    python tests/tools/rocpd_timeline.py <db> w<ms>      every dispatch of the trace's last <ms> milliseconds, game named (multi-game handles)
"""
import re
import sqlite3
import sys


def main(path, steps):
    c = sqlite3.connect(path)
    T = {r[0].rsplit("_0000", 1)[0]: r[0] for r in c.execute("select name from sqlite_master where type='table'")}
    kd, ks = T["rocpd_kernel_dispatch"], T["rocpd_info_kernel_symbol"]
    rows = list(c.execute(f"select d.start, d.end, d.queue_id, s.kernel_name, d.grid_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
    short = lambda n: re.sub(r"^_ZN5pgamd\d+([a-z_0-9]+?)INS_\d+([A-Za-z]+)E(?:Li(\d+)ELi(\d+)E)?.*$", lambda m: m.group(1) + (f"<{m.group(3)}>" if m.group(3) else ""), n)
    if isinstance(steps, str):
        win = float(steps[1:]) * 1e6
        tend = max(r[1] for r in rows)
        sel = [r for r in rows if r[0] >= tend - win]
        t0 = sel[0][0]
        game = lambda n: (re.findall(r"INS_\d+([A-Za-z]+?)(?:TI|E)", n) or ["-"])[0]
        print(f"# {path}: dispatches of the last {steps[1:]} ms; times in us from the first one")
        print("start_us,end_us,dur_us,queue,grid,kernel,game")
        for st, en, q, name, g in sel:
            print(f"{(st - t0) / 1e3:.1f},{(en - t0) / 1e3:.1f},{(en - st) / 1e3:.1f},{q},{g},{short(name)},{game(name)}")
        return
    # a step starts with its first step_tier0 / step_list dispatch after a render dispatch
    starts = [i for i, r in enumerate(rows) if i > 0 and "render" in rows[i - 1][3] and "render" not in r[3]]
    if len(starts) < steps + 1:
        print("too few steps in the trace")
        return
    a, b = starts[-steps - 1], starts[-1]
    t0 = rows[a][0]
    print(f"# {path}: dispatches of {steps} consecutive steps; times in us from the first one")
    print("start_us,end_us,dur_us,queue,grid,kernel")
    for st, en, q, name, g in rows[a:b]:
        print(f"{(st - t0) / 1e3:.1f},{(en - t0) / 1e3:.1f},{(en - st) / 1e3:.1f},{q},{g},{short(name)}")
    step_ms = (rows[b][0] - rows[a][0]) / 1e6 / steps
    print(f"# {step_ms:.3f} ms per step (first dispatch to first dispatch)")


if __name__ == "__main__":
    a = sys.argv[2] if len(sys.argv) > 2 else "3"
    main(sys.argv[1], a if a.startswith("w") else int(a))
