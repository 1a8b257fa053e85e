cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/q3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "peer_mailbox or batch_call or bench_launch" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -30 $O/pytest.log
timeout 400 python bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; python - <<'PY'
import json
j=json.loads(open('gpurun_out/q3/bench_n1.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], {k:v for k,v in j['config'].items() if k!='workload'})
PY
for comm in shm p2p; do
KICP_BENCH_DEVICE=0 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 2 --comm $comm --pg-backend gloo --no-cpu-baseline > $O/bench_2r_$comm.json 2> $O/bench_2r_$comm.err; echo "$comm rc=$?"; tail -c 600 $O/bench_2r_$comm.err; python - <<PY
import json
try:
    j=json.loads(open('gpurun_out/q3/bench_2r_$comm.json').read().strip().splitlines()[-1])
    print("$comm", j['value'], j['config'].get('parallelism'), j['config'].get('multi_iteration',{}).get('scans_per_s'))
except Exception as e: print("no line", e)
PY
done
