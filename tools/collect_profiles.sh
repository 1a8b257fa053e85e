# Round-3 collection run (one gpurun call): GPU tests, the bench line, kernel trace of the same command, PMC passes of the pass
# kernels for cfg2 / cfg5 (generic kernel) and cfg4 (wave-per-query kernel), the other BASELINE workloads, small-scan and pipeline
# timings.  Everything lands under gpurun_out/r03/; the summaries that are meant to be judged are copied into profiles/ by hand
# (profiles/README.md).  KICP_GIT_SHA = the commit the snapshot was taken at (the box has no .git).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" $O/pytest.log | tail -4
timeout 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; tail -c 600 $O/bench_n1.json
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt_bench -o kt -- python bench.py --steps 10 --no-cpu-baseline --no-pmc > $O/bench_under_rocprofv3.json 2> $O/kt_bench.err
python tools/prof_summary.py $(find $O/kt_bench -name "*.db" | head -1) > $O/kernel_trace_stats.txt 2>&1; head -5 $O/kernel_trace_stats.txt
for w in cfg2 cfg5 cfg4; do
  kern=k_pass_gather32; [ $w = cfg4 ] && kern=k_pass_wave
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
KICP_AQL=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_cfg2_hip -o kt -- python tools/prof_target.py --workload cfg2 --calls 300 > /dev/null 2> $O/kt_cfg2_hip.err
python tools/prof_summary.py $(find $O/kt_cfg2_hip -name "*.db" | head -1) > $O/kernel_trace_cfg2_hip_launch.txt 2>&1; grep k_pass $O/kernel_trace_cfg2_hip_launch.txt
for w in cfg1 cfg4 cfg5; do timeout 400 python bench.py --workload $w --cpu-seconds 6 > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$?"; done
timeout 300 python tools/bench_small.py cfg4 > $O/bench_small_cfg4.txt 2>&1; tail -14 $O/bench_small_cfg4.txt | cut -c1-220
timeout 200 python tools/trace_small.py cfg4 > $O/trace_small_cfg4.txt 2>&1
timeout 200 tools/micro/rows > $O/micro_rows.txt 2>&1
timeout 300 python tools/bench_presteps.py > $O/presteps.txt 2>&1; tail -4 $O/presteps.txt
(timeout 300 python tools/bench_pipeline.py --frames 40 --dump /tmp/pipe.bin > /dev/null 2>&1 && timeout 300 tests/cpp/facade_test pipeline_timed /tmp/pipe.bin > /tmp/pipe.txt && timeout 600 python tools/bench_pipeline.py --frames 40 --check /tmp/pipe.txt --oracle-frames 40 2>&1 | grep -v "^frame [0-9]* ms" > $O/pipeline.txt; KICP_TRACE=1 tests/cpp/facade_test pipeline_timed /tmp/pipe.bin 2>&1 >/dev/null | tail -14 > $O/pipeline_calls.txt; timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_pipe -o kt -- tests/cpp/facade_test pipeline_timed /tmp/pipe.bin > /dev/null 2> $O/kt_pipe.err; python tools/prof_summary.py $(find $O/kt_pipe -name "*.db" | head -1) > $O/pipeline_kernel_trace.txt 2>&1); head -3 $O/pipeline.txt
for comm in shm p2p; do
  KICP_BENCH_DEVICE=0 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 2 --comm $comm --pg-backend gloo --no-cpu-baseline --no-pmc > $O/bench_2ranks_1gpu_$comm.json 2> $O/bench_2ranks_1gpu_$comm.err; echo "2 ranks / 1 GPU, $comm: rc=$?"
done
KICP_AQL=0 timeout 400 python bench.py --no-cpu-baseline --no-pmc > $O/bench_n1_hip_launch.json 2> $O/bench_n1_hip_launch.err; echo "hip-launch bench rc=$?"
KICP_KERNARG=host timeout 400 python bench.py --no-cpu-baseline --no-pmc > $O/bench_n1_host_kernarg.json 2> $O/bench_n1_host_kernarg.err; echo "host-kernarg bench rc=$?"
timeout 300 python tools/bench_mapupdate.py > $O/mapupdate.txt 2>&1; tail -2 $O/mapupdate.txt
timeout 300 python tools/gpu_dbg.py cfg2 > $O/ablation_cfg2.txt 2>&1; timeout 300 python tools/gpu_dbg.py cfg5 > $O/ablation_cfg5.txt 2>&1
find $O -name "*.db" -delete
du -sh $O
