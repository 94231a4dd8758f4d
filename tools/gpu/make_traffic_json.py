"""Turns the per-kernel PMC sums of tools/gpu/r3_final.sh (gpurun_out/<tag>_pmc_{FETCH_SIZE,WRITE_SIZE,TCC_HIT_sum}.csv) into
profiles/<out>.json: HBM bytes per step of the default workload.  usage: make_traffic_json.py <tag> <out.json> [num_envs]"""
import json, sys

tag, out = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
rows = {}
for f in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum"):
    for line in open(f"gpurun_out/{tag}_pmc_{f}.csv"):
        p = line.strip().split(",")
        if len(p) == 4 and p[1] in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"):
            rows[(p[0], p[1])] = (int(p[2]), float(p[3]))


def kern(sub):
    return [k for (k, c) in rows if sub in k][0]


def per_launch(sub, c):
    s, v = rows[(kern(sub), c)]
    return v / s


launches = 2  # two env chunks per step: step_tier0 and render are launched once per chunk
r = {c: per_launch("6render", c) * launches for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum")}
s0 = {c: per_launch("step_tier0", c) * launches for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum")}
l1 = {c: per_launch("Li160", c) for c in ("FETCH_SIZE", "WRITE_SIZE")}
l2 = {c: per_launch("Li384", c) for c in ("FETCH_SIZE", "WRITE_SIZE")}
kb = 1024.0
raw = (r["FETCH_SIZE"] + r["WRITE_SIZE"] + s0["FETCH_SIZE"] + s0["WRITE_SIZE"] + l1["FETCH_SIZE"] + l1["WRITE_SIZE"] + l2["FETCH_SIZE"] + l2["WRITE_SIZE"]) * kb
upper = raw + (r["FETCH_SIZE"] + s0["FETCH_SIZE"] + l1["FETCH_SIZE"] + l2["FETCH_SIZE"]) * kb
doc = {
    "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_HIT_sum TCC_MISS_sum (separate passes, tools/gpu/r5_final.sh: the default bench.py command, i.e. averaged over its 1500-step pre-rollout and the timed steps), coinrun num_envs={n}; raw per-kernel sums in the <tag>_pmc_*.csv files (profiles/); computed by tools/gpu/make_traffic_json.py",
    "units": "KB (rocprofv3 FETCH_SIZE / WRITE_SIZE are in KB) per STEP = both chunk launches of step_tier0 and render; list kernels per launch",
    "render_per_step": {"FETCH_SIZE_KB": r["FETCH_SIZE"], "WRITE_SIZE_KB": r["WRITE_SIZE"], "TCC_HIT": r["TCC_HIT_sum"], "TCC_MISS": r["TCC_MISS_sum"]},
    "step_tier0_per_step": {"FETCH_SIZE_KB": s0["FETCH_SIZE"], "WRITE_SIZE_KB": s0["WRITE_SIZE"], "TCC_HIT": s0["TCC_HIT_sum"], "TCC_MISS": s0["TCC_MISS_sum"]},
    "step_list_tier1_per_launch": {"FETCH_SIZE_KB": l1["FETCH_SIZE"], "WRITE_SIZE_KB": l1["WRITE_SIZE"]},
    "step_list_tier2_per_launch": {"FETCH_SIZE_KB": l2["FETCH_SIZE"], "WRITE_SIZE_KB": l2["WRITE_SIZE"]},
    "num_envs": n,
    "calibration": f"render WRITE_SIZE per step = {r['WRITE_SIZE']:.1f} KB vs {n} envs x 12288 B = {n * 12288 / 1024:.1f} KB (the observation write is the only store of that kernel): WRITE_SIZE is exact for this pattern; FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request) as an upper bound for the gather traffic",
    "render_l2_hit_rate": r["TCC_HIT_sum"] / (r["TCC_HIT_sum"] + r["TCC_MISS_sum"]),
    "hbm_bytes_per_step_raw": raw,
    "hbm_bytes_per_step_upper": upper,
    "algorithmic_bytes_per_step": n * 12306,
    "ratio_raw": raw / (n * 12306), "ratio_upper": upper / (n * 12306),
}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps({k: doc[k] for k in ("render_l2_hit_rate", "hbm_bytes_per_step_raw", "hbm_bytes_per_step_upper", "ratio_raw", "ratio_upper")}))
