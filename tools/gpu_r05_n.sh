# round 5: batch_threads default 8 (as many resident kernels as fit the device): tests, per-process probes, cfg1 / cfg4 bench lines
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05n; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_edge.py tests/test_gpu_small.py tests/test_gpu_ties.py tests/test_gpu_configs.py -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log | grep -v "version\|Hostname\|Librccl"
for w in cfg1 cfg4 cfg2; do
  for set in batch_threads=0 ""; do
    timeout 200 python tools/probe_batch.py --workload $w $set 2>&1 | grep "^{"
    timeout 200 python tools/probe_batch.py --workload $w --multi --calls 512 $set 2>&1 | grep "^{"
  done
done | tee $O/probe_default.txt | cut -c1-400
for w in cfg1 cfg4; do timeout 400 python bench.py --workload $w --cpu-seconds 6 --scans 16 --no-pmc > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$?"; done
python - <<'PY'
import json
for w in ("cfg1", "cfg4"):
    d = json.load(open("gpurun_out/r05n/bench_%s.json" % w))
    print(w, {k: d.get(k) for k in ("value", "value_one_scan_in_flight", "ms_per_step")}, d["value_multi_iteration"], d["config"]["scans_in_flight"], (d["cpu_baseline"].get("throughput") or {}).get("scans_per_s"))
PY
