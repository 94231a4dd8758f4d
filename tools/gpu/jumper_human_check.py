"""jumper under render_human on the device against tests/golden/render_human.npz (compiled reference): frames of the three jumper keys.
No pytest / torch: python tools/gpu/jumper_human_check.py"""
import os, sys, zlib
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from procgen_amd import ProcgenGym3Env
gold = np.load(os.path.join(REPO, "tests", "golden", "render_human.npz"))
ok = True
for key, kw in (("jumper", {}), ("jumper@1", dict(distribution_mode="easy")), ("jumper@2", dict(distribution_mode="memory", center_agent=False))):
    env = ProcgenGym3Env(2, "jumper", rand_seed=7, render_mode="rgb_array", **kw)
    acts = gold[f"{key}/actions"]; want = gold[f"{key}/crc"]; k = 0
    for t in range(41):
        env.observe()
        if t in (0, 17, 40):
            rgb = env.info_arrays()["rgb"]
            for e in range(2):
                same = zlib.crc32(rgb[e].tobytes()) == int(want[k][e])
                if t == 17 and e == 0 and f"frames/{key}" in gold.files:
                    d = np.abs(rgb[e].astype(int) - gold[f"frames/{key}"].astype(int))
                    print(key, "full frame: differing pixels", int(np.count_nonzero(d.max(axis=2))), "worst", int(d.max()), flush=True)
                ok &= same
                print(key, "step", t, "env", e, "OK" if same else "DIFFERS", flush=True)
            k += 1
        if t < 40:
            env.act(acts[t])
    st = env.get_state()[0] == gold[f"{key}/state"].tobytes()
    print(key, "state bytes", "OK" if st else "DIFFER", flush=True)
    ok &= st
    env.close()
print("ALL OK" if ok else "FAILED")
