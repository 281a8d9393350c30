# separate --pmc passes (one run each) over tools/prof_cast.py; summaries via tools/rocprof_pmc.py
set -e
export MDP_RC_BATCH=${MDP_RC_BATCH:-1}
P=gpurun_out/prof_cast
rm -rf $P; mkdir -p $P
i=0
for c in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
         "TCC_REQ_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum" \
         "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
         "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" \
         "SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $c -d $P/p$i -o p$i -- python3 tools/prof_cast.py > $P/p$i.log 2>&1
  python3 tools/rocprof_pmc.py $(find $P/p$i -name "*_results.db" | head -1) 2>&1 | grep -E "k_raycaster_update_down|k_post_physics" >> $P/summary.txt
done
find $P -name "*.db" -delete
cat $P/summary.txt
