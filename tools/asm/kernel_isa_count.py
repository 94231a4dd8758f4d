"""Static instruction mix of one kernel of a hipcc object / fat binary: python tools/asm/kernel_isa_count.py file.o raster
(counts by class, and the SGPR spill traffic: v_writelane / v_readlane)."""
import re, subprocess, sys, os, tempfile, collections
path, flt = sys.argv[1], sys.argv[2]
data = open(path, "rb").read()
pos = 0
tmpd = tempfile.mkdtemp()
k = 0
while True:
    pos = data.find(b"\x7fELF", pos)
    if pos < 0:
        break
    if data[pos + 18 : pos + 20] == b"\xe0\x00":
        shoff = int.from_bytes(data[pos + 40 : pos + 48], "little")
        shentsize = int.from_bytes(data[pos + 58 : pos + 60], "little")
        shnum = int.from_bytes(data[pos + 60 : pos + 62], "little")
        size = shoff + shentsize * shnum
        co = os.path.join(tmpd, f"co{k}.o")
        open(co, "wb").write(data[pos : pos + size])
        k += 1
        txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
        cur = None
        cnt = collections.defaultdict(collections.Counter)
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1)
                continue
            if cur is None or flt not in cur or cur.startswith("L") or ".kd" in cur:
                continue
            ins = line.strip().split()
            if not ins or ins[0].startswith("//"):
                continue
            op = ins[0]
            c = cnt[cur]
            c["total"] += 1
            if op.startswith("v_readlane") or op.startswith("v_writelane"):
                c[op.split("_b32")[0]] += 1
            cls = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith(("s_load", "s_waitcnt", "s_nop", "s_buffer")) else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
            c[cls] += 1
        for name, c in cnt.items():
            print(re.sub(r"^_ZN5pgamd\d+", "", name)[:60], dict(c))
        pos += size
    else:
        pos += 4
