"""Per-kernel resources (VGPRs, SGPR spills are not in the metadata, LDS, scratch) of every code object inside a fat binary
(libenv.so): python tools/asm/kernel_table.py procgen_amd/csrc/build/libenv.so [name-filter]"""
import re, subprocess, sys, os, tempfile

so = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
data = open(so, "rb").read()
# code objects are ELF images embedded after "__CLANG_OFFLOAD_BUNDLE__" headers; find every AMDGPU ELF by magic + e_machine 224
out = []
pos = 0
tmpd = tempfile.mkdtemp()
k = 0
while True:
    pos = data.find(b"\x7fELF", pos)
    if pos < 0:
        break
    if data[pos + 18 : pos + 20] == b"\xe0\x00":  # EM_AMDGPU
        # section header offset + count * size gives the image length
        shoff = int.from_bytes(data[pos + 40 : pos + 48], "little")
        shentsize = int.from_bytes(data[pos + 58 : pos + 60], "little")
        shnum = int.from_bytes(data[pos + 60 : pos + 62], "little")
        size = shoff + shentsize * shnum
        path = os.path.join(tmpd, f"co{k}.o")
        open(path, "wb").write(data[pos : pos + size])
        k += 1
        txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", path], capture_output=True, text=True).stdout
        for blk in txt.split("  - .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            g = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", blk).group(1))
            out.append((name, g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
        pos += size
    else:
        pos += 4
for name, v, s, lds, scr in sorted(out):
    short = re.sub(r"^_ZN5pgamd\d+", "", name)
    if flt in name:
        print(f"{short:<70} vgpr {v:>3} sgpr {s:>3} lds {lds:>6} scratch {scr}")
