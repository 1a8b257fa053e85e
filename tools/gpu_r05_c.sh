# round 5, third GPU call: sharded batches with scans in flight (multi-process tests on the one GPU), bench.py under torch.distributed.run
# with two ranks sharing the GPU (shm, now the default exchange behind `value`), the soak
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_shm.py tests/test_gpu_multirank.py -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log | grep -v "version\|Hostname\|Librccl"
KICP_BENCH_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 2 --pg-backend gloo --no-cpu-baseline --no-pmc --scans 16 > $O/bench_2ranks_1gpu_shm.json 2> $O/bench_2ranks_1gpu_shm.err; echo "2 ranks / 1 GPU, shm: rc=$?"
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05c/bench_2ranks_1gpu_shm.json"))
    print({k: d.get(k) for k in ("value", "n_gpus", "value_shm", "value_p2p", "value_rccl", "ms_per_step")}, d["config"].get("parallelism"), d["config"].get("scans_in_flight"))
    print(json.dumps(d["config"].get("exchanges"))[:1500])
    print(d.get("value_replicas"))
except Exception as e:
    print("no line:", e)
PY
tail -5 $O/bench_2ranks_1gpu_shm.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/soak_two_ranks.py --cycles 40 --device 0 2>/dev/null | grep "^{" > $O/soak_two_ranks.txt; cat $O/soak_two_ranks.txt | cut -c1-400
timeout 400 python bench.py --no-cpu-baseline --no-pmc --scans 16 > $O/bench_n1_quick.json 2> $O/bench_n1_quick.err; echo "bench n1 rc=$?"
du -sh $O
