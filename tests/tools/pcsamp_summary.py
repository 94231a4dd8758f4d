"""Per-instruction table of a rocprofv3 PC-sampling run (--pc-sampling-beta-enabled ... --output-format csv).

    python tests/tools/pcsamp_summary.py <output dir> [kernel-name filter] [top N]

Reads <dir>/**/*_pc_sampling_{stochastic,host_trap}.csv (columns as rocprofiler-sdk's tool writes them: Sample_Timestamp, Exec_Mask,
Dispatch_Id, Instruction, Instruction_Comment, Correlation_Id and, for the stochastic method, Wave_Issued_Instruction, Instruction_Type,
Stall_Reason, Wave_Count) and the kernel trace beside it (Dispatch_Id -> Kernel_Name), keeps the samples of dispatches whose kernel
name contains the filter, and prints: samples per kernel; for the filtered kernel the share of samples per stall reason / instruction
type; the top N instructions by samples with their issued / stalled split and dominant stall reason.  The raw CSV stays on the box (it
can be hundreds of MB); this summary is what goes to profiles/.
"""
import collections
import csv
import glob
import os
import sys

csv.field_size_limit(1 << 30)


def main():
    root = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else "render"
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 120
    samples = sorted(glob.glob(os.path.join(root, "**", "*pc_sampling_*.csv"), recursive=True))
    traces = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))
    if not samples:
        print("no *_pc_sampling_*.csv under", root)
        print("files:", [os.path.relpath(p, root) for p in glob.glob(os.path.join(root, "**", "*"), recursive=True)][:40])
        return 1
    kname = {}
    for t in traces:
        with open(t, newline="") as f:
            for row in csv.DictReader(f):
                did = row.get("Dispatch_Id") or row.get("dispatch_id")
                if did is not None:
                    kname[did] = row.get("Kernel_Name", "?")
    print(f"# {len(samples)} sample file(s), {len(kname)} dispatches in the kernel trace; filter = {flt!r}")
    per_kernel = collections.Counter()
    per_inst = {}
    reasons = collections.Counter()
    types = collections.Counter()
    issued_tot = collections.Counter()
    header = None
    n = 0
    for s in samples:
        print("#", os.path.relpath(s, root), os.path.getsize(s) >> 20, "MiB")
        with open(s, newline="") as f:
            rd = csv.DictReader(f)
            header = rd.fieldnames
            for row in rd:
                n += 1
                k = kname.get(row.get("Dispatch_Id", ""), "?")
                short = k.split("(")[0][-70:]
                per_kernel[short] += 1
                if flt not in k:
                    continue
                inst = row.get("Instruction", "?")
                com = row.get("Instruction_Comment", "")
                key = (inst, com)
                e = per_inst.setdefault(key, [0, 0, collections.Counter()])
                e[0] += 1
                issued = row.get("Wave_Issued_Instruction")
                reason = row.get("Stall_Reason", "")
                if issued is not None:
                    if issued.strip() in ("1", "true", "True"):
                        e[1] += 1
                        issued_tot["issued"] += 1
                    else:
                        issued_tot["not issued"] += 1
                        e[2][reason] += 1
                        reasons[reason] += 1
                types[row.get("Instruction_Type", "")] += 1
    print("# columns:", header)
    print(f"# {n} samples")
    print("\n== samples per kernel ==")
    for k, c in per_kernel.most_common(12):
        print(f"{c:9d} {100.0 * c / max(n, 1):6.2f} %  {k}")
    tot = sum(e[0] for e in per_inst.values())
    print(f"\n== kernels matching {flt!r}: {tot} samples ==")
    if issued_tot:
        print("issued / not issued:", dict(issued_tot))
        print("not-issued reasons:")
        for r, c in reasons.most_common():
            print(f"  {c:9d} {100.0 * c / max(tot, 1):6.2f} %  {r}")
        print("instruction type at the sampled pc:")
        for r, c in types.most_common():
            print(f"  {c:9d} {100.0 * c / max(tot, 1):6.2f} %  {r}")
    # by opcode
    by_op = collections.Counter()
    for (inst, _), e in per_inst.items():
        by_op[inst.split(" ")[0]] += e[0]
    print("\n== by opcode ==")
    for op, c in by_op.most_common(40):
        print(f"{c:9d} {100.0 * c / max(tot, 1):6.2f} %  {op}")
    print(f"\n== top {top} instructions (samples, share, issued, dominant stall reason, instruction, source comment) ==")
    for (inst, com), e in sorted(per_inst.items(), key=lambda kv: -kv[1][0])[:top]:
        dom = e[2].most_common(1)
        dom_s = f"{dom[0][0].replace('ROCPROFILER_PC_SAMPLING_INSTRUCTION_NOT_ISSUED_REASON_', '')}:{dom[0][1]}" if dom else "-"
        print(f"{e[0]:8d} {100.0 * e[0] / max(tot, 1):6.2f} %  iss {e[1]:6d}  {dom_s:<28} {inst[:70]:<70} {com[-60:]}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
