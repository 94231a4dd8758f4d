"""
Build-container tool (needs oracle/_ref, the compiled reference): random (game, distribution_mode, option subset, seed)
combinations, compiled reference vs oracle vs the emulated kernels, frames / rewards / first flags for 70 steps each.

    python tests/tools/fuzz_options.py [master seed]

Rounds with master seeds 1-3 (192 combinations) had no mismatch at the end of round 1; its deterministic cousins are
tests/golden/mode_matrix.npz and option_matrix.npz.
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, os.path.join(REPO, "tests", "emu"))
import ref_env, oracle_env, emu_harness
MODES={"easy":0,"hard":1,"extreme":2,"memory":10}
EXT={"chaser","dodgeball","leaper","starpilot"}; MEM={"caveflyer","dodgeball","heist","jumper","maze","miner"}
ALL=["bigfish","bossfight","caveflyer","chaser","climber","coinrun","dodgeball","fruitbot","heist","jumper","leaper","maze","miner","ninja","plunder","starpilot"]
master=np.random.RandomState(int(sys.argv[1]) if len(sys.argv)>1 else 0)
N=5; T=70; total=0; fails=0
for game in ALL:
    for rep in range(4):
        modes=["easy","hard"]+(["extreme"] if game in EXT else [])+(["memory"] if game in MEM else [])
        mode=modes[master.randint(len(modes))]
        kw={}
        for name in ("use_backgrounds","center_agent"):
            if master.rand()<0.4: kw[name]=False
        for name in ("restrict_themes","use_monochrome_assets","paint_vel_info","use_sequential_levels"):
            if master.rand()<0.4: kw[name]=True
        if master.rand()<0.5: kw["num_levels"]=int(master.randint(1,5)); kw["start_level"]=int(master.randint(0,1000))
        if kw.get("use_sequential_levels") and "num_levels" not in kw: kw["num_levels"]=2
        seed=int(master.randint(0,2**31-1))
        rk=dict(kw); rk["distribution_mode"]=mode; ok=dict(kw); ok["distribution_mode"]=MODES[mode]
        ref = ref_env.make_ref_env(N, game, rand_seed=seed, **rk); orc = oracle_env.OracleEnv(N, game, rand_seed=seed, **ok); emu = emu_harness.EmuEnv(N, game, rand_seed=seed, **ok)
        rng=np.random.RandomState(seed%1000); bad=0; bad2=0
        for t in range(T):
            r1,o1,f1 = ref.observe(); r2,o2,f2 = orc.observe(); r3,o3,f3 = emu.observe()
            if not (np.array_equal(r1,r2) and np.array_equal(f1,f2) and np.array_equal(o1['rgb'],o2['rgb'])): bad+=1
            if not (np.array_equal(r3,r2) and np.array_equal(f3,f2) and np.array_equal(o3['rgb'],o2['rgb'])): bad2+=1
            a=rng.randint(0,15,size=(N,),dtype=np.int32); ref.act(a); orc.act(a); emu.act(a)
        ref.close(); total+=1
        if bad or bad2:
            fails+=1; print("MISMATCH", game, mode, kw, "seed", seed, "oracle-vs-ref", bad, "emu-vs-oracle", bad2, flush=True)
print("combos", total, "failing", fails)
