"""Round 4 tooling: VGPR liveness over the AMDGPU assembly of one kernel (hipcc -S): where is the register-pressure peak, and which
values are live there (with the line that defined each)?  Approximate: every VGPR write is taken as a full definition (a write
under a partial exec mask does not really kill the old value), v_writelane as use + def.
  python tools/asm/vgpr_liveness.py kernel.s [function-substring] [top-N peaks]"""
import re, sys

path = sys.argv[1]
fsub = sys.argv[2] if len(sys.argv) > 2 else ""
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_][\w$.]*:", l) and fsub in l and not l.startswith(".L"))
end = next(i for i in range(start, len(lines)) if ".Lfunc_end" in lines[i])
body = lines[start + 1 : end]


def vregs(tok):
    out = []
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return [int(m.group(1))]
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


NO_DST = ("global_store", "buffer_store", "ds_write", "flat_store", "scratch_store", "s_", "v_cmp", "v_cmpx", "ds_bpermute_dummy", "global_atomic", "v_readlane", "v_readfirstlane", "ds_append", "buffer_wbl2", "global_load_lds")
insts = []  # (line_no, op, defs, uses, targets, falls)
labels = {}
for i, l in enumerate(body):
    t = l.split(";")[0].strip()
    if not t:
        continue
    m = re.match(r"^(\.L\w+):", t)
    if m:
        labels[m.group(1)] = len(insts)
        continue
    if t.startswith("."):
        continue
    parts = t.split(None, 1)
    op = parts[0]
    ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", parts[1])] if len(parts) > 1 else []
    ops = [o.split()[0] if o else o for o in ops]  # drop modifiers like "offset:4"
    defs, uses = [], []
    if op.startswith("global_atomic") or op.startswith("ds_") and ("rtn" in op or op.startswith("ds_read") or op.startswith("ds_bpermute") or op.startswith("ds_swizzle")):
        if ops:
            defs += vregs(ops[0])
        for o in ops[1:]:
            uses += vregs(o)
    elif any(op.startswith(p) for p in NO_DST):
        for o in ops:
            uses += vregs(o)
    else:
        if ops:
            defs += vregs(ops[0])
        for o in ops[1:]:
            uses += vregs(o)
        if op.startswith("v_writelane") or "sdwa" in op and "dst_unused:UNUSED_PRESERVE" in l or op.startswith("v_mac") or op.startswith("v_fmac") or op.startswith("v_pk_fmac"):
            uses += defs
    targets, falls = [], True
    if op in ("s_branch",):
        targets, falls = [ops[0]], False
    elif op.startswith("s_cbranch"):
        targets = [ops[0]]
    elif op in ("s_endpgm",):
        falls = False
    insts.append((i, op, defs, uses, targets, falls))

n = len(insts)
succ = [[] for _ in range(n)]
for k, (_, op, _, _, targets, falls) in enumerate(insts):
    if falls and k + 1 < n:
        succ[k].append(k + 1)
    for t in targets:
        if t in labels and labels[t] < n:
            succ[k].append(labels[t])
live_in = [0] * n
defm = [sum(1 << r for r in set(d)) for (_, _, d, _, _, _) in insts]
usem = [sum(1 << r for r in set(u)) for (_, _, _, u, _, _) in insts]
changed = True
it = 0
while changed:
    changed = False
    it += 1
    for k in range(n - 1, -1, -1):
        out = 0
        for s_ in succ[k]:
            out |= live_in[s_]
        new = usem[k] | (out & ~defm[k])
        if new != live_in[k]:
            live_in[k] = new
            changed = True
cnt = [bin(x).count("1") for x in live_in]
order = sorted(range(n), key=lambda k: -cnt[k])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 1
print(f"{n} instructions, {it} passes; max live VGPRs {cnt[order[0]]}")
# last definition line of each register before point k (program order approximation)
seen = []
for k in order:
    if any(abs(k - s) < 400 for s in seen):
        continue
    seen.append(k)
    print(f"\n== peak at asm line {start + 2 + insts[k][0]} ({insts[k][1]}), {cnt[k]} live")
    regs = [r for r in range(512) if (live_in[k] >> r) & 1]
    info = []
    for r in regs:
        dl = None
        for j in range(k - 1, -1, -1):
            if r in insts[j][2]:
                dl = j
                break
        nu = None
        for j in range(k, n):
            if r in insts[j][3]:
                nu = j
                break
        info.append((r, dl, nu))
    for r, dl, nu in sorted(info, key=lambda x: (x[1] if x[1] is not None else -1)):
        dtxt = f"{start + 2 + insts[dl][0]}:{insts[dl][1]}" if dl is not None else "entry"
        utxt = f"{start + 2 + insts[nu][0]}:{insts[nu][1]}" if nu is not None else "-"
        print(f"  v{r:<3} def {dtxt:<40} next use {utxt}")
    if len(seen) >= top:
        break
