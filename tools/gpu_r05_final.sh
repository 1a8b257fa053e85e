# round 5, at the round's last code commit: the whole GPU suite, smoke(), the bench lines of every BASELINE workload (cfg2 and cfg5 with the
# in-run PMC passes), two ranks on one GPU
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05final; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" $O/pytest.log | tail -8 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?"; grep smoke $O/smoke.log | cut -c1-200
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
for w in cfg1 cfg4; do timeout 400 python bench.py --workload $w --cpu-seconds 6 --scans 16 --no-pmc > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$?"; done
timeout 500 python bench.py --workload cfg5 --cpu-seconds 6 --scans 16 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 rc=$?"
KICP_BENCH_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 2 --pg-backend gloo --no-cpu-baseline --no-pmc --scans 16 > $O/bench_2ranks_1gpu_shm.json 2> $O/bench_2ranks_1gpu_shm.err; echo "2 ranks / 1 GPU, shm: rc=$?"
python - <<'PY'
import json
for w in ("n1", "cfg1", "cfg4", "cfg5"):
    d = json.load(open("gpurun_out/r05final/bench_%s.json" % w))
    c = d["cpu_baseline"]
    print(w, d["value"], d.get("value_one_scan_in_flight"), d["value_multi_iteration"]["scans_per_s"], d["value_multi_iteration"]["us_per_iteration"],
          "cpu", c.get("value"), c.get("cores"), c.get("single_thread_value"), (c.get("throughput") or {}).get("scans_per_s"))
d = json.load(open("gpurun_out/r05final/bench_2ranks_1gpu_shm.json"))
print("2 ranks", d["value"], d["config"]["exchanges"]["shm"], d.get("value_p2p"))
PY
du -sh $O
