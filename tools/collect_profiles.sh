# Round-3 collection run, final state of the round (one gpurun call): GPU tests, the bench line, kernel trace of the same command,
# PMC passes of the generic pass kernel (cfg2 in full, cfg5 traffic + wave-cycle counters), the other BASELINE workloads, small-scan,
# pipeline, concurrency, in-process A/B and exchange timings.  Everything lands under gpurun_out/r03/; the summaries that are meant
# to be judged are copied into profiles/ by hand (profiles/README.md says which commit each file was taken at).
# KICP_GIT_SHA = the commit the snapshot was taken at (the box has no .git).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" $O/pytest.log | tail -4
timeout 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; tail -c 600 $O/bench_n1.json
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_bench -o kt -- python bench.py --steps 10 --no-cpu-baseline --no-pmc > $O/bench_under_rocprofv3.json 2> $O/kt_bench.err
python tools/prof_summary.py $(find $O/kt_bench -name "*.db" | head -1) > $O/kernel_trace_stats.txt 2>&1; head -5 $O/kernel_trace_stats.txt
for w in cfg2 cfg5; do
  kern=k_pass_gather32
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$w -o kt -- python tools/prof_target.py --workload $w --calls 300 > $O/kt_$w.json 2> $O/kt_$w.err
  python tools/prof_summary.py $(find $O/kt_$w -name "*.db" | head -1) > $O/kernel_trace_$w.txt 2>&1; grep k_pass $O/kernel_trace_$w.txt
  i=0
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" "VALUBusy" "MeanOccupancyPerCU"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $c -d $O/pmc_${w}_$i -o pmc -- python tools/prof_target.py --workload $w --calls 200 > /dev/null 2> $O/pmc_${w}_$i.err || echo "pmc pass $i ($c) failed for $w"
  done
  avg=$(grep $kern $O/kernel_trace_$w.txt | head -1 | awk '{print $(NF-3)}')
  python tools/prof_counters_json.py $O/r03_counters_$w.json $kern ${avg:-0} $(find $O/pmc_${w}_* -name "*.db") > $O/counters_$w.txt 2>&1; cat $O/counters_$w.txt | cut -c1-400
done
for w in cfg1 cfg4 cfg5; do timeout 400 python bench.py --workload $w --cpu-seconds 6 > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$?"; done
timeout 300 python tools/bench_small.py cfg4 > $O/bench_small_cfg4.txt 2>&1; tail -14 $O/bench_small_cfg4.txt | cut -c1-220
timeout 200 python tools/trace_small.py cfg4 > $O/trace_small_cfg4.txt 2>&1
(timeout 200 python tools/bench_concurrent.py --workload cfg2 --lanes 1 2 3 4 6 8; timeout 200 python tools/bench_concurrent.py --workload cfg2 --multi --count 128 --lanes 2 4; timeout 200 python tools/bench_concurrent.py --workload cfg4 --lanes 2 4 8; timeout 200 python tools/bench_concurrent.py --workload cfg5 --count 128 --lanes 2 4) 2>&1 | grep "^{" > $O/bench_concurrent.txt; cut -c1-300 $O/bench_concurrent.txt
(timeout 200 python tools/ab_option.py --workload cfg2 --sets base latency_kernel=0 base latency_kernel=0; timeout 200 python tools/ab_option.py --workload cfg2 --multi --calls 50 --sets base latency_kernel=0 resident_generic=0 small=0 base latency_kernel=0 resident_generic=0 small=0; timeout 200 python tools/ab_option.py --workload cfg1 --multi --calls 100 --sets base resident_generic=0 base resident_generic=0; timeout 200 python tools/ab_option.py --workload cfg5 --calls 60 --sets base latency_kernel=2 base latency_kernel=2) 2>&1 | grep "^{" > $O/ab_options.txt; cut -c1-400 $O/ab_options.txt
timeout 100 python tools/time_presteps.py 2>&1 | grep "^{" > $O/time_presteps.txt; KICP_DIRECT_UPLOAD=1 timeout 100 python tools/time_presteps.py 2>&1 | grep "^{" >> $O/time_presteps.txt
timeout 300 python tools/bench_presteps.py > $O/presteps.txt 2>&1; tail -4 $O/presteps.txt
(timeout 300 python tools/bench_pipeline.py --frames 40 --dump /tmp/pipe.bin > /dev/null 2>&1 && timeout 300 tests/cpp/facade_test pipeline_timed /tmp/pipe.bin > /tmp/pipe.txt && timeout 600 python tools/bench_pipeline.py --frames 40 --check /tmp/pipe.txt --oracle-frames 40 2>&1 | grep -v "^frame [0-9]* ms" > $O/pipeline.txt; KICP_TRACE=1 tests/cpp/facade_test pipeline_timed /tmp/pipe.bin 2>&1 >/dev/null | tail -14 > $O/pipeline_calls.txt; timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_pipe -o kt -- tests/cpp/facade_test pipeline_timed /tmp/pipe.bin > /dev/null 2> $O/kt_pipe.err; python tools/prof_summary.py $(find $O/kt_pipe -name "*.db" | head -1) > $O/pipeline_kernel_trace.txt 2>&1); head -3 $O/pipeline.txt
for comm in shm p2p; do
  KICP_BENCH_DEVICE=0 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 2 --comm $comm --pg-backend gloo --no-cpu-baseline --no-pmc > $O/bench_2ranks_1gpu_$comm.json 2> $O/bench_2ranks_1gpu_$comm.err; echo "2 ranks / 1 GPU, $comm: rc=$?"
done
for v in 1 0; do KICP_P2P_ROWS=$v timeout 300 python bench.py --force-comm --comm p2p --pg-backend gloo --no-cpu-baseline --no-pmc --steps 20 2>/dev/null | tail -1 > $O/bench_1rank_p2p_rows$v.json; done
timeout 300 python tools/bench_mapupdate.py > $O/mapupdate.txt 2>&1; tail -2 $O/mapupdate.txt
find $O -name "*.db" -delete
du -sh $O
