"""Per-kernel wait / issue table from the CSVs of tools/gpu/stall_counters.sh (rows `kernel,counter,samples,sum` as tests/tools/rocpd_summary.py prints them).
SQ counters are sampled per shader engine / XCC and summed over the dispatch; every figure below is a RATIO of such sums, so the unit drops out:
  wave-cycles per wave, share of wave-cycles spent waiting (any s_waitcnt / dependency) and waiting for an instruction slot, share of
  wave-cycles in which an instruction of each type was being issued, average latency of a VMEM / SMEM / LDS instruction
  (accumulated in-flight level / instructions issued), LDS bank-conflict cycles per LDS-active cycle, texture-addresser busy and stall shares."""
import collections, re, sys

val = collections.defaultdict(dict)
for path in sys.argv[1:]:
    for ln in open(path):
        p = ln.rstrip("\n").split(",")
        if len(p) == 4 and p[1] not in ("counter",) and re.match(r"^[A-Z]", p[1]):
            try:
                val[p[0]][p[1]] = float(p[3])
            except ValueError:
                pass


def short(k):
    m = re.match(r"_ZN5pgamd\d+(\w+?)INS_\d+(\w+?)E", k)
    return f"{m.group(1)}<{m.group(2)}>" if m else k[:40]


def ratio(d, a, b, scale=1.0):
    return f"{scale * d[a] / d[b]:10.3f}" if a in d and b in d and d[b] else "         -"


rows = [
    ("wave-cycles per wave", "SQ_WAVE_CYCLES", "SQ_WAVES", 1),
    ("waiting (any) / wave-cycles", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", 1),
    ("waiting for issue / wave-cycles", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", 1),
    ("waiting on LDS instr / wave-cycles", "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES", 1),
    ("issuing any / wave-cycles", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES", 1),
    ("issuing VALU / wave-cycles", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", 1),
    ("issuing scalar / wave-cycles", "SQ_ACTIVE_INST_SCA", "SQ_WAVE_CYCLES", 1),
    ("issuing LDS / wave-cycles", "SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES", 1),
    ("issuing VMEM / wave-cycles", "SQ_ACTIVE_INST_VMEM", "SQ_WAVE_CYCLES", 1),
    ("issuing FLAT / wave-cycles", "SQ_ACTIVE_INST_FLAT", "SQ_WAVE_CYCLES", 1),
    ("busy cycles x4 SIMDs / wave-cycles (1 / waves per SIMD resident)", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", 4),
    ("VALU instr per wave", "SQ_INSTS_VALU", "SQ_WAVES", 1),
    ("SALU instr per wave", "SQ_INSTS_SALU", "SQ_WAVES", 1),
    ("VMEM reads per wave", "SQ_INSTS_VMEM_RD", "SQ_WAVES", 1),
    ("VMEM writes per wave", "SQ_INSTS_VMEM_WR", "SQ_WAVES", 1),
    ("SMEM per wave", "SQ_INSTS_SMEM", "SQ_WAVES", 1),
    ("LDS instr per wave", "SQ_INSTS_LDS", "SQ_WAVES", 1),
    ("avg VMEM latency (cycles)", "SQ_INST_LEVEL_VMEM", "SQ_INSTS_VMEM_RD", 1),
    ("avg SMEM latency (cycles)", "SQ_INST_LEVEL_SMEM", "SQ_INSTS_SMEM", 1),
    ("avg LDS latency (cycles)", "SQ_INST_LEVEL_LDS", "SQ_INSTS_LDS", 1),
    ("active lanes per VALU instr", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", 1),
    ("LDS bank conflict / LDS active", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", 1),
    ("TA addr stalled by TC / TA busy", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_TA_BUSY_sum", 1),
    ("TA data stalled by TC / TA busy", "TA_DATA_STALLED_BY_TC_CYCLES_sum", "TA_TA_BUSY_sum", 1),
    ("TCP pending stall / TA busy", "TCP_PENDING_STALL_CYCLES_sum", "TA_TA_BUSY_sum", 1),
    ("TCP->TCC read latency (cycles per request)", "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum", 1),
]
kernels = [k for k in val if "render" in k or "step_" in k]
kernels.sort(key=lambda k: -val[k].get("SQ_WAVE_CYCLES", 0))
# SQ_WAVES etc. come from different passes: merge is per kernel name, so ratios across passes assume the same launches (same command)
print(f"{'':66}" + "".join(f"{short(k)[:22]:>24}" for k in kernels))
for name, a, b, sc in rows:
    print(f"{name:66}" + "".join(f"{ratio(val[k], a, b, sc):>24}" for k in kernels))
print("\nraw sums:")
for k in kernels:
    print(short(k), {c: v for c, v in sorted(val[k].items())})
