# One parametrised collection script for the GPU box (replaces the per-experiment tools/gpu_r0*_?.sh of earlier rounds):
#     gpurun --timeout 1500 -- 'bash tools/gpu.sh <out-tag> <stage> [<stage> ...]'
# Stages (each writes under gpurun_out/<out-tag>/):
#   pretests   the pre-step / ingest / facade / golden-pipeline GPU tests
#   tests      the whole GPU suite
#   pipeline   the drop-in RegisterFrame on 40 frames of 131 072 points, raw and raw_ahead, 3 runs each + the C-ABI calls' wall times
#   timeline   one frame's kernels, start offsets and queues, from a kernel trace of the drive
#   pipeab     the same drive under sets of environment switches (AB_SETS), with the calls' wall times and their laps
#   pipetrace  rocprofv3 kernel trace of the raw_ahead drive (pre-step / map kernels per frame)
#   bench      bench.py at the default workload (cfg2), the driver's command
#   benchall   bench.py for cfg1, cfg4, cfg5 as well
#   trace      rocprofv3 --kernel-trace --stats of the bench command
#   counters   PMC passes of the pass kernel (cfg2, cfg5)
#   counters4  the same for cfg4's wave-per-query kernel (FETCH_SIZE / WRITE_SIZE / TCC), one launch per pass
#   ranks2     two ranks on one GPU (shm and rccl exchanges)
#   smoke      __graft_entry__.smoke()
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
quiet() { grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl"; }
# the process that drives the GPU runs on one L3 domain of the GPU's NUMA node (tools/bench_pipeline.py placement(); NEAR="" where unknown)
near_gpu() { [ -n "${NEAR_SET:-}" ] || { CPUS=$(python -c "import kinematic_icp_amd as K; print(','.join(map(str, sorted(K.cpus_near_gpu(0)))))" 2>/dev/null); NEAR=""; [ -n "$CPUS" ] && [ "${KICP_BENCH_PLACEMENT:-1}" != 0 ] && NEAR="taskset -c $CPUS"; NEAR_SET=1; echo "caller placement: ${NEAR:-none}"; }; }
pipe_dump() { [ -f /tmp/pipe.bin ] || timeout 300 python tools/bench_pipeline.py --frames 40 --mode raw --dump /tmp/pipe.bin > /dev/null 2>&1; }
for stage in "$@"; do
case $stage in
pretests)
  ( time timeout 900 python -m pytest tests/test_ingest.py tests/test_facade.py tests/test_gpu_presteps.py tests/test_golden_pipeline.py -m gpu -x -q --timeout 120 ) > $O/pretests.log 2>&1
  echo "pretests rc=$?"; tail -12 $O/pretests.log | quiet ;;
tests)
  ( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
  quiet < $O/pytest.log | tail -12 > $O/gpu_tests.txt; cat $O/gpu_tests.txt ;;
pipeline)
  pipe_dump; near_gpu
  for m in raw raw_ahead; do
    mode=pipeline_timed_raw; [ $m = raw_ahead ] && mode=pipeline_timed_raw_ahead
    for rep in 1 2 3; do
      $NEAR timeout 300 tests/cpp/facade_test $mode /tmp/pipe.bin > /tmp/pipe_$m.txt
      timeout 900 python tools/bench_pipeline.py --frames 40 --mode $m --check /tmp/pipe_$m.txt --oracle-frames 0 --ref-frames 0 2>&1 | grep -v "^frame [0-9]* ms" > $O/pipeline_${m}_$rep.txt
      grep "GPU RegisterFrame\|^drive" $O/pipeline_${m}_$rep.txt | cut -c1-220
    done
    KICP_TRACE=1 $NEAR tests/cpp/facade_test $mode /tmp/pipe.bin 2>&1 >/dev/null | grep "^\[kicp" | tail -100 | head -24 > $O/pipeline_calls_$m.txt
    echo "== $m"; cat $O/pipeline_calls_$m.txt
  done ;;
pipetrace)
  pipe_dump; near_gpu
  for m in raw raw_ahead; do
    $NEAR timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_pipe_$m -o kt -- tests/cpp/facade_test pipeline_timed_$m /tmp/pipe.bin > /dev/null 2> $O/kt_pipe_$m.err
    python tools/prof_summary.py $(find $O/kt_pipe_$m -name "*.db" | head -1) > $O/pipeline_kernel_trace_$m.txt 2>&1; echo "== $m"; head -16 $O/pipeline_kernel_trace_$m.txt | cut -c1-160
    counts=$(KICP_TRACE=1 $NEAR tests/cpp/facade_test pipeline_timed_$m /tmp/pipe.bin 2>&1 >/dev/null | grep "chained pre-steps" | tail -1 | sed "s/.*steps: \([0-9]*\) -> \([0-9]*\) -> \([0-9]*\) -> \([0-9]*\) points.*/\1 \2 \3 \4/")
    python tools/pipeline_table.py $O/pipeline_kernel_trace_$m.txt $counts > $O/pipeline_kernel_table_$m.txt 2>&1; cat $O/pipeline_kernel_table_$m.txt | cut -c1-200
    rm -rf $O/kt_pipe_$m
  done ;;
timeline)  # one frame's kernels with start offsets and queues (tools/pipeline_timeline.py), raw and raw_ahead
  pipe_dump; near_gpu
  for m in raw raw_ahead; do
    $NEAR timeout 300 rocprofv3 --kernel-trace -d $O/tl_$m -o kt -- tests/cpp/facade_test pipeline_timed_$m /tmp/pipe.bin > /dev/null 2> $O/tl_$m.err
    python tools/pipeline_timeline.py $(find $O/tl_$m -name "*.db" | head -1) 2>&1 | grep -v "^tables\|^rocpd\|copies; sample" > $O/pipeline_timeline_$m.txt; cat $O/pipeline_timeline_$m.txt | cut -c1-150
    rm -rf $O/tl_$m
  done ;;
pipeab)   # A/B of environment switches on the drop-in frame: AB_SETS="A=1,B=2 A=0 ..." ("-" = the defaults); the sets take turns, AB_REPS rounds
  pipe_dump; near_gpu
  for rep in $(seq 1 ${AB_REPS:-3}); do
    for set in ${AB_SETS:-- KICP_PRE_PUSH_WGS=0}; do
      envs=$(echo $set | tr ',' ' '); [ "$set" = "-" ] && envs="KICP_AB_DEFAULTS=1"
      for m in raw raw_ahead; do
        env $envs $NEAR timeout 300 tests/cpp/facade_test pipeline_timed_$m /tmp/pipe.bin > /tmp/pipe_ab.txt
        echo "$set $m: $(timeout 300 python tools/bench_pipeline.py --frames 40 --mode $m --check /tmp/pipe_ab.txt --oracle-frames 0 --ref-frames 0 2>&1 | grep 'GPU RegisterFrame\|^drive' | sed 's/the last 20 frames in //; s/wall clock.*//; s/(second half.*p10/p10/; s/first.*//' | tr '\n' ' ')"
        [ $rep = 1 ] && env $envs KICP_TRACE=1 $NEAR tests/cpp/facade_test pipeline_timed_$m /tmp/pipe.bin 2>&1 >/dev/null | grep "^\[kicp" | tail -120 | head -40 > $O/pipeline_ab_calls_${set//[^A-Za-z0-9]/_}_$m.txt
      done
    done
  done 2>&1 | sort -s -k1,2 | tee $O/pipeline_ab.txt ;;
bench)
  timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; tail -c 600 $O/bench_n1.json; echo ;;
benchall)
  for w in cfg1 cfg4; do timeout 400 python bench.py --workload $w --cpu-seconds 6 --scans 16 --no-pmc > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$?"; done
  timeout 500 python bench.py --workload cfg5 --cpu-seconds 6 --scans 16 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 rc=$?" ;;
trace)
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_bench -o kt -- python bench.py --steps 10 --no-cpu-baseline --no-pmc --scans 16 > $O/bench_under_rocprofv3.json 2> $O/kt_bench.err
  python tools/prof_summary.py $(find $O/kt_bench -name "*.db" | head -1) > $O/kernel_trace_stats.txt 2>&1; head -12 $O/kernel_trace_stats.txt | cut -c1-160
  rm -rf $O/kt_bench ;;
counters)
  for w in cfg2 cfg5; do
    kern=k_pass_gather32; bt="--batch 64"; calls=256; [ $w = cfg5 ] && calls=96
    timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$w -o kt -- python tools/prof_target.py --workload $w --calls $((2*calls)) $bt > $O/kt_$w.json 2> $O/kt_$w.err
    python tools/prof_summary.py $(find $O/kt_$w -name "*.db" | head -1) > $O/kernel_trace_$w.txt 2>&1; grep k_pass $O/kernel_trace_$w.txt | cut -c1-160
    i=0
    for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
             "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" "VALUBusy" "MeanOccupancyPerCU"; do
      i=$((i+1))
      timeout 150 rocprofv3 --pmc $c -d $O/pmc_${w}_$i -o pmc -- python tools/prof_target.py --workload $w --calls $calls $bt > /dev/null 2> $O/pmc_${w}_$i.err || echo "pmc pass $i ($c) failed for $w"
    done
    avg=$(grep $kern $O/kernel_trace_$w.txt | head -1 | awk '{print $(NF-3)}')
    KICP_GIT_SHA=$(cat .git_sha 2>/dev/null) python tools/prof_counters_json.py $O/counters_$w.json $kern ${avg:-0} $(find $O/pmc_${w}_* -name "*.db") > $O/counters_$w.txt 2>&1; cut -c1-300 $O/counters_$w.txt
    rm -rf $O/kt_$w $O/pmc_${w}_*
  done ;;
counters4)  # cfg4 (1 080-point scans, the wave-per-query kernel): HBM traffic per pass, one launch per pass so that dispatches = passes
  kern=k_pass_wave
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_cfg4 -o kt -- python tools/prof_target.py --workload cfg4 --calls 600 --option small_resident=0 > $O/kt_cfg4.json 2> $O/kt_cfg4.err
  python tools/prof_summary.py $(find $O/kt_cfg4 -name "*.db" | head -1) > $O/kernel_trace_cfg4.txt 2>&1; grep k_pass $O/kernel_trace_cfg4.txt | cut -c1-160
  i=0
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1))
    timeout 150 rocprofv3 --pmc $c -d $O/pmc_cfg4_$i -o pmc -- python tools/prof_target.py --workload cfg4 --calls 300 --option small_resident=0 > /dev/null 2> $O/pmc_cfg4_$i.err || echo "pmc pass $i ($c) failed for cfg4"
  done
  avg=$(grep $kern $O/kernel_trace_cfg4.txt | head -1 | awk '{print $(NF-3)}')
  KICP_GIT_SHA=$(cat .git_sha 2>/dev/null) python tools/prof_counters_json.py $O/counters_cfg4.json $kern ${avg:-0} $(find $O/pmc_cfg4_* -name "*.db") > $O/counters_cfg4.txt 2>&1; cut -c1-300 $O/counters_cfg4.txt
  rm -rf $O/kt_cfg4 $O/pmc_cfg4_* ;;
ranks2)
  for comm in shm rccl; do
    KICP_BENCH_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 2 --comm $comm --pg-backend gloo --no-cpu-baseline --no-pmc --scans 16 > $O/bench_2ranks_1gpu_$comm.json 2> $O/bench_2ranks_1gpu_$comm.err
    echo "2 ranks / 1 GPU, $comm: rc=$?"; tail -c 400 $O/bench_2ranks_1gpu_$comm.json; echo
  done ;;
smoke)
  ( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?"; grep smoke $O/smoke.log | cut -c1-200 ;;
*) echo "unknown stage $stage" ;;
esac
done
find $O -name "*.db" -delete
du -sh $O
