# round 6: why does the launch order by background image cost bigfish a third of its rate (profiles/r05_render_order_ab.txt)?  L2 (TCC) counters with the order off / on
# for bigfish and coinrun: requests, hits, misses, tag stalls, and -- where rocprofv3 exposes the dimension -- per channel.
TAG=${1:-r6tcc}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
rocprofv3 --list-avail 2>/dev/null | grep -i "TCC_\|TCP_TCC\|GRBM_GUI" | cut -c1-160 | head -80 > gpurun_out/${TAG}_avail.txt; wc -l gpurun_out/${TAG}_avail.txt
cd /tmp && export TMPDIR=/tmp
for g in bigfish coinrun; do
  for k in 0 16; do
    for c in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RD_UNCACHED_32B_sum"; do
      n=$(echo $c | cut -d' ' -f1)
      PROCGEN_AMD_RENDER_ORDER=$k timeout 200 rocprofv3 --pmc $c --kernel-trace -d /tmp/${TAG}_${g}_${k}_$n -o p -- python $R/bench.py --game $g --steps 8 --warmup 2 --steady-warmup 300 --no-cpu-baseline --no-host-landed --no-traffic > $R/gpurun_out/${TAG}_${g}_${k}_$n.log 2>&1
      DB=$(find /tmp/${TAG}_${g}_${k}_$n -name "*.db" | head -1)
      [ -n "$DB" ] && python $R/tests/tools/rocpd_summary.py $DB 2>&1 | grep "raster\|6renderI" | cut -c1-170 | sed "s/^/$g order=$k: /" | tee -a $R/gpurun_out/${TAG}_table.txt
      rm -rf /tmp/${TAG}_${g}_${k}_$n
    done
  done
done
