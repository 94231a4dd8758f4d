"""Per-kernel VGPR / SGPR / LDS / scratch and code size of one game's device code (hipcc -S --cuda-device-only).
    python tools/kernel_resources.py CoinRun [extra hipcc flags]"""
import os, re, subprocess, sys, tempfile
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "procgen_amd", "csrc")
game = sys.argv[1]
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-strict-aliasing", f"-DPG_GAME={game}", "--cuda-device-only", "-S", "-c",
                           os.path.join(CSRC, "kernels_game.hip"), "-o", out] + sys.argv[2:], cwd=CSRC, stderr=subprocess.DEVNULL)
    text = open(out).read()
blocks = re.findall(r"- \.agpr_count.*?\.wavefront_size:\s+\d+", text, re.S)
for b in blocks:
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", b).group(1)
    name = re.sub(r"^_ZN5pgamd\d+([a-z_0-9]+?)I.*?(?:Li(\d+)ELi\d+E)?E.*$", lambda m: m.group(1) + (f"<{m.group(2)}>" if m.group(2) else ""), g("name"))
    print(f"{name:18s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>4s}")
for m in re.finditer(r"^; codeLenInByte = (\d+)", text, re.M):
    pass
print("code bytes per kernel:", re.findall(r"; codeLenInByte = (\d+)", text))
