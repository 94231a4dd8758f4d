"""Reads a PROCGEN_AMD_DEBUG=8192 residency trace (the render kernel records its part only in a library built with EXTRA=-DPG_RENDER_TRACE=1)
 (per env: step start / end / kind<<32|HW_ID, render start / end / HW_ID; 100 MHz
ticks) and prints, for the last step: kernel spans, mean workgroup lifetimes, resident workgroups over time, per-CU spread."""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 32)
n = raw.shape[0]
t0 = min(raw[:, 0].min(), raw[:, 4][raw[:, 4] > 0].min())


def us(x):
    return (x.astype(np.int64) - np.int64(t0)) / 100.0


for name, b in (("step", 0), ("render", 4)):
    s, e, hw = us(raw[:, b]), us(raw[:, b + 1]), raw[:, b + 2]
    kind = (hw >> np.uint64(32)).astype(int)
    print(f"== {name}: {n} workgroups, span {s.min():.1f} .. {e.max():.1f} us, mean lifetime {np.mean(e - s):.2f} us (p50 {np.median(e - s):.2f}, p99 {np.percentile(e - s, 99):.2f}, max {np.max(e - s):.2f})")
    for k in np.unique(kind):
        m = kind == k
        print(f"   kind {k}: {m.sum()} envs, span {s[m].min():.1f} .. {e[m].max():.1f}, mean lifetime {np.mean((e - s)[m]):.2f} us")
    # resident count sampled every 20 us
    grid = np.arange(s.min(), e.max(), 20.0)
    res = [(np.sum((s <= t) & (e > t))) for t in grid]
    print("   resident workgroups every 20 us:", " ".join(str(r) for r in res))
    h = (hw & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    # gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ... (xcc id is in XCC_ID register, not here)
    cu = (h >> 8) & 0xF
    se = (h >> 13) & 0x7
    simd = (h >> 4) & 0x3
    print("   per-SIMD share:", np.bincount(simd, minlength=4) / n, " per-CU-id share:", np.round(np.bincount(cu, minlength=16) / n, 3))

# the tail: the workgroups of the step kernels that end last, and the longest ones
s, e, hw = us(raw[:, 0]), us(raw[:, 1]), raw[:, 2]
kind = (hw >> np.uint64(32)).astype(int)
order = np.argsort(-e)[:25]
print("== step workgroups ending last: env kind start end lifetime")
for i in order:
    print(f"   {i:6d} {kind[i]} {s[i]:8.1f} {e[i]:8.1f} {e[i] - s[i]:8.1f}")
order = np.argsort(-(e - s))[:25]
print("== longest step workgroups: env kind start end lifetime")
for i in order:
    print(f"   {i:6d} {kind[i]} {s[i]:8.1f} {e[i]:8.1f} {e[i] - s[i]:8.1f}")
lt = e - s
print("== lifetime histogram (us):", " ".join(f"<{b}:{int(np.sum(lt < b))}" for b in (10, 20, 40, 80, 160, 320, 640, 1280)))
for k in np.unique(kind):
    m = kind == k
    print(f"   kind {k} start-time percentiles (us): p0 {s[m].min():.1f} p50 {np.median(s[m]):.1f} p90 {np.percentile(s[m], 90):.1f} p100 {s[m].max():.1f}; total wave-time {np.sum(lt[m]) / 1e3:.1f} ms")
first = raw[:, 3].astype(int)
print(f"== episodes that ended in this step: {first.sum()}; their lifetimes: mean {lt[first == 1].mean():.1f} us, p50 {np.median(lt[first == 1]):.1f}, p90 {np.percentile(lt[first == 1], 90):.1f}, max {lt[first == 1].max():.1f}")
print(f"   workgroups above 300 us: {int(np.sum(lt > 300))}, of them ended episodes: {int(np.sum((lt > 300) & (first == 1)))}")
print(f"   lifetimes of ended episodes, histogram: ", " ".join(f"<{b}:{int(np.sum(lt[first == 1] < b))}" for b in (40, 80, 160, 320, 640, 1280)))
if raw[:, 8:24].sum() > 0:
    names = ["load_env", "action+velocity", "step_entities (rest)", "collision_pass", "erase_if_needed", "game_step tail", "reset", "outputs+camera", "store_env", "bso: setup", "bso: sub_steps", "se: find+plain", "se: smart ent_step", "13", "14", "15"]
    ph = raw[:, 8:24].astype(np.float64)
    long_ = lt > 300
    print("== phase cycles: mean over all envs | mean over the workgroups above 300 us")
    for k in range(16):
        if ph[:, k].sum() > 0:
            print(f"   {names[k]:22s} {ph[:, k].mean():10.0f} | {ph[long_, k].mean():10.0f}")
    cn = ["sub_steps (serial path)", "entity_scan<0> calls", "push_obj calls (any depth)", "cycles in entity_scan<0>"]
    c = raw[:, 24:28].astype(np.float64)
    print("== counters: mean over all envs | mean over the workgroups above 300 us | max")
    for k in range(4):
        print(f"   {cn[k]:28s} {c[:, k].mean():12.2f} | {c[long_, k].mean():12.2f} | {c[:, k].max():12.0f}")
    print("   share of envs with a push:", np.mean(c[:, 2] > 0), " of long ones:", np.mean(c[long_, 2] > 0), "; long ones among envs with a push:", np.mean(long_[c[:, 2] > 0]))
    print("   share of envs with an entity scan:", np.mean(c[:, 1] > 0), " of long ones:", np.mean(c[long_, 1] > 0), "; long ones among them:", np.mean(long_[c[:, 1] > 0]))
