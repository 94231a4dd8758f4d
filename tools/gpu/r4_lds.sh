# round 4: cell table as bytes (render LDS 17.3 -> 14.5 KB / 11.1 -> 8.3 KB): same-box A/B against the round-3 library
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
python tools/gpu/ab_bench.py tools/gpu/ab/libenv_r03.so,procgen_amd/csrc/build/libenv.so coinrun,maze,chaser,heist,caveflyer,leaper,dodgeball,fruitbot,miner,ninja 2>&1 | awk 'NR%2==0' | tee gpurun_out/r4_lds.txt
