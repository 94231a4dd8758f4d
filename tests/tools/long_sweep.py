"""Build-container tool (needs oracle/_ref): the emulated kernels against the COMPILED REFERENCE over longer horizons than the fixtures hold -- 19 game /
mode / option cases x 8 envs x 400 steps with forced resets, frames + rewards + first flags every step and the get_state bytes at the end.
    python tests/tools/long_sweep.py [seed0=90925777] [envs=8] [steps=400]"""
import os, sys, time
REPO=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (REPO, REPO+'/oracle', REPO+'/tests', REPO+'/tests/emu'): sys.path.insert(0,p)
import numpy as np, emu_harness, ref_env
MODES={"easy":0,"hard":1,"extreme":2,"memory":10}
cases=[("coinrun","hard",{}),("coinrun","easy",{}),("bossfight","hard",{}),("fruitbot","hard",{}),("dodgeball","extreme",{}),("leaper","extreme",{}),("starpilot","extreme",{}),("caveflyer","memory",{}),("heist","memory",{}),("chaser","extreme",{}),("maze","memory",{}),("jumper","hard",{}),("plunder","hard",{}),("ninja","hard",{}),("climber","hard",{}),("miner","memory",{}),("bigfish","hard",{}),("fruitbot","easy",{"center_agent":False}),("coinrun","hard",{"center_agent":False,"use_backgrounds":False,"paint_vel_info":True})]
seed0=int(sys.argv[1]) if len(sys.argv)>1 else 90925777
N_ENVS=int(sys.argv[2]) if len(sys.argv)>2 else 8
N_STEPS=int(sys.argv[3]) if len(sys.argv)>3 else 400
tot=0
for k,(game,mode,kw) in enumerate(cases):
    n,steps=N_ENVS,N_STEPS
    ref=ref_env.make_ref_env(n,game,rand_seed=seed0+k,distribution_mode=mode,**kw)
    emu=emu_harness.EmuEnv(n,game,rand_seed=seed0+k,distribution_mode=MODES[mode],**kw)
    rng=np.random.RandomState((seed0+k)%(2**31))
    bad=0
    for t in range(steps+1):
        r1,o1,f1=ref.observe(); r2,o2,f2=emu.observe()
        if not (np.array_equal(r1,r2) and np.array_equal(np.asarray(f1).astype(bool),np.asarray(f2).astype(bool)) and np.array_equal(o1['rgb'],o2['rgb'])): bad+=1
        if t<steps:
            a=rng.randint(0,15,size=n).astype(np.int32)
            if t%97==96: a[rng.randint(n)]=-1
            ref.act(a); emu.act(a)
    st_ok = ref.get_state()==emu.get_state()
    print(game,mode,kw,'mismatching steps',bad,'state bytes equal',st_ok,flush=True)
    tot+=bad+(0 if st_ok else 1)
    ref.close(); emu.close()
print('TOTAL',tot)
