# round 5, fourth GPU call: the small-scan kernels hand their sums over through group accumulators (tests + in-process A/B on cfg4 and on
# pipeline-sized scans), the hypothesis fuzz
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_small.py tests/test_gpu_edge.py tests/test_gpu_ties.py tests/test_gpu_fuzz.py -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log | grep -v "version\|Hostname\|Librccl"
A="timeout 300 python tools/ab_option.py"
( $A --workload cfg4 --calls 400 --blocks 24 --option small_group_rows --values 1 0
  $A --workload cfg4 --calls 200 --blocks 16 --multi --option small_group_rows --values 1 0
  $A --workload cfg4 --batch --calls 1024 --blocks 16 --option small_group_rows --values 1 0
  $A --workload cfg4 --batch --calls 512 --blocks 12 --multi --option small_group_rows --values 1 0
  $A --workload cfg4 --batch --calls 1024 --blocks 12 --fixed small_group_rows=1 --sets base batch_depth=4 batch_depth=2
  $A --workload cfg4 --calls 400 --blocks 16 --fixed small_wave=0 --option small_group_rows --values 1 0 ) 2>&1 | grep "^{" | tee $O/ab_small_group_rows.txt | cut -c1-400
timeout 400 python bench.py --workload cfg4 --cpu-seconds 6 --scans 16 --no-pmc > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "cfg4 rc=$?"; python - <<'PY'
import json
d = json.load(open("gpurun_out/r05d/bench_cfg4.json"))
print({k: d.get(k) for k in ("value", "value_one_scan_in_flight", "ms_per_step")}, d["value_multi_iteration"], d["cpu_baseline"].get("value"), d["cpu_baseline"].get("throughput"))
PY
du -sh $O
