# What the waves of the render / step kernels wait on, from hardware counters (rocprofv3 PC sampling is refused on this box: profiles/r05_pc_sampling_unavailable.txt).
# Five separate --pmc passes of the steady-state bench (one counter set per pass, --kernel-trace only), summed per kernel by tests/tools/rocpd_summary.py;
# tools/gpu/stall_table.py turns the CSVs into the per-kernel table.   usage: bash tools/gpu/stall_counters.sh [tag]
TAG=${1:-r5_stall}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_THREAD_CYCLES_VALU" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/${TAG}_p$i -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${TAG}_pass$i.log 2>&1
  python $R/tests/tools/rocpd_summary.py $(find /tmp/${TAG}_p$i -name "*.db" | head -1) > $R/gpurun_out/${TAG}_pass$i.csv 2>&1
  rm -rf /tmp/${TAG}_p$i
  grep -c "render" $R/gpurun_out/${TAG}_pass$i.csv
done
python $R/tools/gpu/stall_table.py $R/gpurun_out/${TAG}_pass*.csv | tee $R/gpurun_out/${TAG}_table.txt
