# Round-4 collection run, final state of the round (one gpurun call): GPU tests, the bench line, kernel trace of the same command,
# PMC passes of the generic pass kernel (cfg2 in full, cfg5 traffic + wave-cycle counters), the other BASELINE workloads, the
# resident-pass timeline, the ablation, pipeline (PointCloud2 records and fp64 vectors), two ranks on one GPU (bench lines + soak),
# in-process A/Bs.  Everything lands under gpurun_out/r04/; the summaries that are meant to be judged are copied into profiles/
# (profiles/README.md says which commit each file was taken at).  KICP_GIT_SHA = the commit the snapshot was taken at.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" $O/pytest.log | tail -4
timeout 500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; tail -c 400 $O/bench_n1.json
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt_bench -o kt -- python bench.py --steps 10 --no-cpu-baseline --no-pmc --scans 16 > $O/bench_under_rocprofv3.json 2> $O/kt_bench.err
python tools/prof_summary.py $(find $O/kt_bench -name "*.db" | head -1) > $O/kernel_trace_stats.txt 2>&1; head -6 $O/kernel_trace_stats.txt
for w in cfg2 cfg5; do
  kern=k_pass_gather32
  bt=""; [ $w = cfg2 ] && bt="--batch 64"   # cfg2: through batch calls, as bench.py times it (four scans in flight, the four-waves-per-SIMD build)
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$w -o kt -- python tools/prof_target.py --workload $w --calls 300 $bt > $O/kt_$w.json 2> $O/kt_$w.err
  python tools/prof_summary.py $(find $O/kt_$w -name "*.db" | head -1) > $O/kernel_trace_$w.txt 2>&1; grep k_pass $O/kernel_trace_$w.txt
  i=0
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" "VALUBusy" "MeanOccupancyPerCU"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $c -d $O/pmc_${w}_$i -o pmc -- python tools/prof_target.py --workload $w --calls 200 $bt > /dev/null 2> $O/pmc_${w}_$i.err || echo "pmc pass $i ($c) failed for $w"
  done
  avg=$(grep $kern $O/kernel_trace_$w.txt | head -1 | awk '{print $(NF-3)}')
  python tools/prof_counters_json.py $O/r04_counters_$w.json $kern ${avg:-0} $(find $O/pmc_${w}_* -name "*.db") > $O/counters_$w.txt 2>&1; cat $O/counters_$w.txt | cut -c1-300
done
for w in cfg1 cfg4 cfg5; do timeout 400 python bench.py --workload $w --cpu-seconds 6 --scans 16 > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$?"; done
for w in cfg2 cfg5; do timeout 200 python tools/gpu_dbg.py $w; done > $O/ablation.txt 2>&1; tail -16 $O/ablation.txt
timeout 300 python tools/trace_resident.py cfg2 > $O/trace_resident_cfg2.txt 2>&1; timeout 300 python tools/trace_resident.py cfg1 > $O/trace_resident_cfg1.txt 2>&1; tail -9 $O/trace_resident_cfg2.txt
timeout 200 python tools/trace_small.py cfg4 > $O/trace_small_cfg4.txt 2>&1
(for w in cfg2 cfg1 cfg4; do for m in "" "--multi"; do timeout 300 python tools/ab_option.py --workload $w --batch $m --calls 256 --blocks 24 --sets batch_queues=0,batch_resident=0 batch_queues=0,batch_depth=1 batch_queues=0,batch_depth=2,batch_rotate=0 batch_queues=0,batch_depth=2 batch_queues=0 base; done; done) 2>&1 | grep "^{" > $O/ab_batch_modes.txt; cut -c1-600 $O/ab_batch_modes.txt
timeout 300 python tools/trace_batch.py cfg2 --depths 1 2 3 > $O/trace_batch_cfg2.txt 2>&1; timeout 200 python tools/trace_batch.py cfg4 --depths 1 3 > $O/trace_batch_cfg4.txt 2>&1; grep "per workgroup\|batch_depth" $O/trace_batch_cfg2.txt
(timeout 200 python tools/bench_concurrent.py --workload cfg2 --lanes 1 2 4; timeout 200 python tools/bench_concurrent.py --workload cfg4 --lanes 2 4) 2>&1 | grep "^{" > $O/bench_concurrent.txt; cut -c1-300 $O/bench_concurrent.txt
python - > $O/latency_probe.txt 2>&1 <<'PY'
import kinematic_icp_amd as K
print("kicp_probe_dependent_load (ns per dependent step; every lane chases its own chain through random 128-byte lines)")
for ws in (16 << 10, 2 << 20, 16 << 20, 74 << 20, 256 << 20, 1 << 30):
    print("  working set %8.1f MB: one wave per CU %7.1f ns | 512 x 256 lanes (cfg2's launch shape) %7.1f ns | 1954 x 256 lanes (cfg5's) %7.1f ns" % (
        ws / 2 ** 20, K.probe_dependent_load(ws, 256, 64, 256), K.probe_dependent_load(ws, 512, 256, 64), K.probe_dependent_load(ws, 1954, 256, 32)))
PY
cat $O/latency_probe.txt
for m in raw vectors; do
  mode=pipeline_timed; [ $m = raw ] && mode=pipeline_timed_raw
  (timeout 300 python tools/bench_pipeline.py --frames 40 --mode $m --dump /tmp/pipe_$m.bin > /dev/null 2>&1 && timeout 300 tests/cpp/facade_test $mode /tmp/pipe_$m.bin > /tmp/pipe_$m.txt && timeout 600 python tools/bench_pipeline.py --frames 40 --mode $m --check /tmp/pipe_$m.txt --oracle-frames 40 2>&1 | grep -v "^frame [0-9]* ms" > $O/pipeline_$m.txt; KICP_TRACE=1 tests/cpp/facade_test $mode /tmp/pipe_$m.bin 2>&1 >/dev/null | tail -12 > $O/pipeline_calls_$m.txt); head -3 $O/pipeline_$m.txt
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_pipe -o kt -- tests/cpp/facade_test pipeline_timed_raw /tmp/pipe_raw.bin > /dev/null 2> $O/kt_pipe.err; python tools/prof_summary.py $(find $O/kt_pipe -name "*.db" | head -1) > $O/pipeline_kernel_trace.txt 2>&1
KICP_ROCTX=1 timeout 300 rocprofv3 --marker-trace --kernel-trace --stats -d $O/kt_roctx -o kt -- tests/cpp/facade_test pipeline_timed_raw /tmp/pipe_raw.bin > /dev/null 2> $O/kt_roctx.err; (find $O/kt_roctx -name "*marker*stats*" -o -name "*stats*.csv" | head -3; python tools/prof_markers.py $(find $O/kt_roctx -name "*.db" | head -1)) > $O/pipeline_roctx_ranges.txt 2>&1; head -12 $O/pipeline_roctx_ranges.txt
for comm in shm p2p; do
  KICP_BENCH_DEVICE=0 timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 2 --comm $comm --pg-backend gloo --no-cpu-baseline --no-pmc --scans 16 > $O/bench_2ranks_1gpu_$comm.json 2> $O/bench_2ranks_1gpu_$comm.err; echo "2 ranks / 1 GPU, $comm: rc=$?"
done
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/soak_two_ranks.py --cycles 200 --device 0 2>/dev/null | grep "^{" > $O/soak_two_ranks.txt; cat $O/soak_two_ranks.txt
timeout 300 python tools/bench_mapupdate.py > $O/mapupdate.txt 2>&1; tail -2 $O/mapupdate.txt
find $O -name "*.db" -delete; rm -rf $O/kt_* $O/pmc_*
du -sh $O
