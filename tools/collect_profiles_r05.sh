# Round 5 collection (one gpurun call): the whole GPU suite, the bench lines of every BASELINE workload at this commit (cfg2 with the in-run
# PMC passes: traffic AND SQ_INSTS_VALU), the kernel trace of the bench command, PMC counters of the pass kernel as the timed region runs it
# (cfg2, cfg5), the pipeline with the reference's own RegisterFrame timed beside it, two ranks on one GPU, smoke.  Everything lands under
# gpurun_out/r05/; the summaries meant to be judged are copied into profiles/ (profiles/README.md names the commit).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" $O/pytest.log | tail -8 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; tail -c 300 $O/bench_n1.json; echo
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_bench -o kt -- python bench.py --steps 10 --no-cpu-baseline --no-pmc --scans 16 > $O/bench_under_rocprofv3.json 2> $O/kt_bench.err
python tools/prof_summary.py $(find $O/kt_bench -name "*.db" | head -1) > $O/kernel_trace_stats.txt 2>&1; head -8 $O/kernel_trace_stats.txt
for w in cfg2 cfg5; do
  kern=k_pass_gather32; bt="--batch 64"; calls=256; [ $w = cfg5 ] && calls=96
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$w -o kt -- python tools/prof_target.py --workload $w --calls $((2*calls)) $bt > $O/kt_$w.json 2> $O/kt_$w.err
  python tools/prof_summary.py $(find $O/kt_$w -name "*.db" | head -1) > $O/kernel_trace_$w.txt 2>&1; grep k_pass $O/kernel_trace_$w.txt
  i=0
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" "VALUBusy" "MeanOccupancyPerCU"; do
    i=$((i+1))
    timeout 150 rocprofv3 --pmc $c -d $O/pmc_${w}_$i -o pmc -- python tools/prof_target.py --workload $w --calls $calls $bt > /dev/null 2> $O/pmc_${w}_$i.err || echo "pmc pass $i ($c) failed for $w"
  done
  avg=$(grep $kern $O/kernel_trace_$w.txt | head -1 | awk '{print $(NF-3)}')
  KICP_GIT_SHA=$(cat .git_sha 2>/dev/null) python tools/prof_counters_json.py $O/r05_counters_$w.json $kern ${avg:-0} $(find $O/pmc_${w}_* -name "*.db") > $O/counters_$w.txt 2>&1; cat $O/counters_$w.txt | cut -c1-300
done
for w in cfg1 cfg4; do timeout 400 python bench.py --workload $w --cpu-seconds 6 --scans 16 --no-pmc > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$?"; done
timeout 500 python bench.py --workload cfg5 --cpu-seconds 6 --scans 16 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 rc=$?"
for m in raw; do
  mode=pipeline_timed_raw
  (timeout 300 python tools/bench_pipeline.py --frames 40 --mode $m --dump /tmp/pipe_$m.bin > /dev/null 2>&1 && timeout 300 tests/cpp/facade_test $mode /tmp/pipe_$m.bin > /tmp/pipe_$m.txt && timeout 900 python tools/bench_pipeline.py --frames 40 --mode $m --check /tmp/pipe_$m.txt --oracle-frames 0 --ref-frames 40 2>&1 | grep -v "^frame [0-9]* ms" > $O/pipeline_$m.txt; KICP_TRACE=1 tests/cpp/facade_test $mode /tmp/pipe_$m.bin 2>&1 >/dev/null | tail -14 > $O/pipeline_calls_$m.txt); tail -4 $O/pipeline_$m.txt | cut -c1-300
done
KICP_BENCH_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 2 --pg-backend gloo --no-cpu-baseline --no-pmc --scans 16 > $O/bench_2ranks_1gpu_shm.json 2> $O/bench_2ranks_1gpu_shm.err; echo "2 ranks / 1 GPU, shm: rc=$?"
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -4 $O/smoke.log
find $O -name "*.db" -delete; rm -rf $O/kt_* $O/pmc_*
du -sh $O
