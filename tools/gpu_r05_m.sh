# round 5: cfg1-size scans on several resident kernels side by side, one setting per process (tools/probe_batch.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05m; mkdir -p $O
for set in batch_threads=0 batch_threads=3 batch_threads=4 batch_threads=6 batch_threads=8 "batch_threads=0 batch_queues=0"; do
  timeout 200 python tools/probe_batch.py --workload cfg1 $set 2>&1 | grep "^{"
  timeout 200 python tools/probe_batch.py --workload cfg1 --multi --calls 512 $set 2>&1 | grep "^{"
done | tee $O/probe_cfg1.txt | cut -c1-400
