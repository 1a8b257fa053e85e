cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/q7; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
timeout 400 python bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; python - <<'PY'
import json
j=json.loads(open('gpurun_out/q7/bench_n1.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], {k:v for k,v in j['config'].items() if k!='workload'})
print({k:v for k,v in j['roofline'].items() if k not in ('counters','note')})
PY
for w in cfg1 cfg4 cfg5; do timeout 400 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$?"; python - <<PY
import json
j=json.loads(open('gpurun_out/q7/bench_$w.json').read().strip().splitlines()[-1])
print("$w", j['value'], j['config']['ms_per_scan'], j['config']['multi_iteration']['scans_per_s'], j['roofline']['kernel_avg_us'])
PY
done
