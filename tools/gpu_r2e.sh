cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2e; mkdir -p $O
for d in 0 3 5; do
  i=0
  for c in "TCP_GATE_EN1_sum TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_MULTI_MISS_sum" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES" \
           "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $c -d $O/pmc_${d}_$i -o pmc -- python tools/prof_target.py --workload cfg2 --calls 200 --option dbg=$d > /dev/null 2> $O/pmc_${d}_$i.err || echo "pass $i failed (dbg $d)"
    python tools/prof_counters.py $(find $O/pmc_${d}_$i -name "*.db" | head -1) 2>&1 | grep -i "k_pass" | sed "s/^/dbg$d /"
  done
done
find $O -name "*.db" -delete
