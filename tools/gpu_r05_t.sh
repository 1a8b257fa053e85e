# round 5: bench.py with two ranks on ONE GPU, cfg1 (8 192-point shards: the sharded batch goes through resident kernels side by side)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05t; mkdir -p $O
export KICP_WAIT_TIMEOUT_S=20
( time timeout 600 python -m pytest tests/test_gpu_shm.py tests/test_gpu_multirank.py -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log | grep -v "version\|Hostname\|Librccl"
for bt in 3 0; do
KICP_BENCH_DEVICE=0 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2968$bt bench.py --gpus 2 --workload cfg1 --pg-backend gloo --no-cpu-baseline --no-pmc --scans 16 --no-sharded-cfg5 --set shard_threads=$bt > $O/bench_2ranks_cfg1_bt$bt.json 2> $O/bench_2ranks_cfg1_bt$bt.err; echo "bt=$bt rc=$?"
python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r05t/bench_2ranks_cfg1_bt$bt.json"))
    print(d["value"], d["config"].get("scans_in_flight"), d["config"].get("scans_per_step"), json.dumps(d["config"]["exchanges"].get("shm"))[:400])
except Exception as e:
    print("no line:", e)
PY
done
grep -h 'kicp error' $O/*.err | sort | uniq -c | cut -c1-200
