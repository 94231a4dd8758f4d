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


# Per STEP, not per launch: a run may hold launches of several sizes (the two uneven chunks; the four chunks of a host-landed handle), so the
# steps are counted by what the render kernels wrote -- WRITE_SIZE is exact for the observation store (12288 B per env-frame) -- and
# every sum is divided by that.
def total(sub, c):
    return rows[(kern(sub), c)][1]


steps = total("6render", "WRITE_SIZE") * 1024.0 / (n * 12288.0)
r = {c: total("6render", c) / steps for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum")}
s0 = {c: total("step_tier0", c) / steps for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum")}
l1 = {c: total("Li160", c) / steps for c in ("FETCH_SIZE", "WRITE_SIZE")}
l2 = {c: total("Li384", c) / steps for c in ("FETCH_SIZE", "WRITE_SIZE")}
kb = 1024.0
raw = (r["FETCH_SIZE"] + r["WRITE_SIZE"] + s0["FETCH_SIZE"] + s0["WRITE_SIZE"] + l1["FETCH_SIZE"] + l1["WRITE_SIZE"] + l2["FETCH_SIZE"] + l2["WRITE_SIZE"]) * kb
upper = raw + (r["FETCH_SIZE"] + s0["FETCH_SIZE"] + l1["FETCH_SIZE"] + l2["FETCH_SIZE"]) * kb
doc = {
    "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_HIT_sum TCC_MISS_sum (separate passes, tools/gpu/r5_final.sh: the default bench.py command, i.e. averaged over its 1500-step pre-rollout and the timed steps), coinrun num_envs={n}; raw per-kernel sums in the <tag>_pmc_*.csv files (profiles/); computed by tools/gpu/make_traffic_json.py",
    "units": "KB (rocprofv3 FETCH_SIZE / WRITE_SIZE are in KB) per STEP: a kernel's sum over the run divided by the steps the run rendered (render WRITE_SIZE / (num_envs x 12288 B))", "steps_in_run": steps,
    "render_per_step": {"FETCH_SIZE_KB": r["FETCH_SIZE"], "WRITE_SIZE_KB": r["WRITE_SIZE"], "TCC_HIT": r["TCC_HIT_sum"], "TCC_MISS": r["TCC_MISS_sum"]},
    "step_tier0_per_step": {"FETCH_SIZE_KB": s0["FETCH_SIZE"], "WRITE_SIZE_KB": s0["WRITE_SIZE"], "TCC_HIT": s0["TCC_HIT_sum"], "TCC_MISS": s0["TCC_MISS_sum"]},
    "step_list_tier1_per_step": {"FETCH_SIZE_KB": l1["FETCH_SIZE"], "WRITE_SIZE_KB": l1["WRITE_SIZE"]},
    "step_list_tier2_per_step": {"FETCH_SIZE_KB": l2["FETCH_SIZE"], "WRITE_SIZE_KB": l2["WRITE_SIZE"]},
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
