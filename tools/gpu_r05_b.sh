# round 5, second GPU call: the counting accumulators on the ordinary launch (parity tests, in-process A/B against round 4's hand-over,
# dbg 14), how many queues the batch call wants now that a pass has fewer instructions, a bench line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_ties.py tests/test_gpu_configs.py tests/test_gpu_ranges.py tests/test_gpu_small.py tests/test_gpu_shm.py -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log | grep -v "version\|Hostname\|Librccl"
A="timeout 300 python tools/ab_option.py"
( $A --workload cfg2 --batch --calls 512 --blocks 16 --option dbg --values 0 14
  $A --workload cfg2 --batch --calls 256 --blocks 12 --multi --option dbg --values 0 14
  $A --workload cfg2 --calls 250 --blocks 24 --option dbg --values 0 14
  $A --workload cfg2 --calls 60 --blocks 16 --multi --option dbg --values 0 14
  $A --workload cfg1 --batch --calls 512 --blocks 12 --option dbg --values 0 14
  $A --workload cfg5 --batch --calls 128 --blocks 8 --option dbg --values 0 14
  $A --workload cfg5 --calls 60 --blocks 12 --option dbg --values 0 14 ) 2>&1 | grep "^{" | tee $O/ab_handover.txt | cut -c1-400
( $A --workload cfg2 --batch --calls 768 --blocks 12 --sets batch_queues=3 base batch_queues=5 batch_queues=6 batch_queues=8
  $A --workload cfg2 --batch --calls 384 --blocks 8 --multi --sets base batch_queues=5 batch_queues=6 batch_queues=8
  $A --workload cfg5 --batch --calls 128 --blocks 8 --sets batch_queues=2 batch_queues=3 base batch_queues=6 ) 2>&1 | grep "^{" | tee $O/ab_queues.txt | cut -c1-500
timeout 500 python bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; python - <<'PY'
import json
d = json.load(open("gpurun_out/r05b/bench_n1.json"))
print({k: d.get(k) for k in ("value", "value_one_scan_in_flight", "ms_per_step")}, d["value_multi_iteration"], d["roofline"]["time_split_us"], d["roofline"]["kernel_avg_us"])
PY
du -sh $O
