# GPU session A of round 2: tests, bench, micro-benchmarks, kernel trace + PMC passes (cfg2 and cfg5).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2a; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 3000 $O/bench.json
timeout 200 tools/micro/handoff > $O/handoff.txt 2>&1; cat $O/handoff.txt
timeout 200 python tools/prof_target.py --workload cfg2 --calls 2000 --events > $O/target_cfg2_dev.json 2> $O/target.err; cat $O/target_cfg2_dev.json
timeout 300 python tools/prof_target.py --workload cfg2 --calls 2000 --events --build host > $O/target_cfg2_host.json 2>> $O/target.err; cat $O/target_cfg2_host.json
timeout 300 python tools/prof_target.py --workload cfg2 --calls 1000 --events --multi > $O/target_cfg2_multi.json 2>> $O/target.err; cat $O/target_cfg2_multi.json
timeout 300 python tools/prof_target.py --workload cfg5 --calls 500 --events > $O/target_cfg5.json 2>> $O/target.err; cat $O/target_cfg5.json
rocprofv3 -L > $O/counters_list.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt_bench -o kt -- python bench.py --steps 10 --no-cpu-baseline > $O/bench_under_rocprofv3.json 2> $O/kt_bench.err
python tools/prof_summary.py $(find $O/kt_bench -name "*.db" | head -1) > $O/kernel_trace_bench.txt 2>&1; head -6 $O/kernel_trace_bench.txt
for w in cfg2 cfg5; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$w -o kt -- python tools/prof_target.py --workload $w --calls 300 > $O/kt_$w.json 2> $O/kt_$w.err
  python tools/prof_summary.py $(find $O/kt_$w -name "*.db" | head -1) > $O/kernel_trace_$w.txt 2>&1; head -4 $O/kernel_trace_$w.txt
  i=0
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "VALUBusy" "MeanOccupancyPerCU" "MemUnitBusy" "TA_BUSY_avr TA_TA_BUSY_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $c -d $O/pmc_${w}_$i -o pmc -- python tools/prof_target.py --workload $w --calls 300 > /dev/null 2> $O/pmc_${w}_$i.err || echo "pmc pass $i ($c) failed for $w"
  done
  avg=$(python - <<PY
import re
for l in open("$O/kernel_trace_$w.txt"):
    if "k_pass_gather32" in l:
        print(l.split()[-4]); break
PY
)
  python tools/prof_counters_json.py $O/r02_counters_$w.json k_pass_gather32 ${avg:-0} $(find $O/pmc_${w}_* -name "*.db") > $O/counters_$w.txt 2>&1; cat $O/counters_$w.txt
done
find $O -name "*.db" -delete
du -sh $O
