# round 5: batches of small scans on several resident kernels side by side - tests, in-process A/B on cfg4, a cfg4 bench line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05i; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_edge.py tests/test_gpu_small.py -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log | grep -v "version\|Hostname\|Librccl"
A="timeout 300 python tools/ab_option.py"
( $A --workload cfg4 --batch --calls 1024 --blocks 12 --sets batch_threads=0 batch_threads=2 base batch_threads=4
  $A --workload cfg4 --batch --calls 512 --blocks 10 --multi --sets batch_threads=0 batch_threads=2 base batch_threads=4
  $A --workload cfg4 --batch --calls 1024 --blocks 10 --fixed small_group_rows=2 --sets batch_threads=0 batch_threads=2 base batch_threads=4 ) 2>&1 | grep "^{" | tee $O/ab_batch_threads.txt | cut -c1-400
timeout 400 python bench.py --workload cfg4 --cpu-seconds 6 --scans 16 --no-pmc > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "cfg4 rc=$?"; python - <<'PY'
import json
d = json.load(open("gpurun_out/r05i/bench_cfg4.json"))
print({k: d.get(k) for k in ("value", "value_one_scan_in_flight", "ms_per_step")}, d["value_multi_iteration"], d["config"]["scans_in_flight"], (d["cpu_baseline"].get("throughput") or {}).get("scans_per_s"))
PY
