# round 5: large scans (cfg2: 512 workgroups) on resident kernels of the FOUR-WAVES build - one, and two side by side - against the four queues
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05q; mkdir -p $O
for set in "" "batch_threads_large=1" "batch_threads=0 batch_queues=0" "batch_threads=0 batch_queues=0 resident_four_waves=1" "batch_threads_large=1 batch_depth=2" "batch_threads_large=1 batch_depth=4"; do
  timeout 200 python tools/probe_batch.py --workload cfg2 $set 2>&1 | grep "^{"
  timeout 200 python tools/probe_batch.py --workload cfg2 --multi --calls 512 $set 2>&1 | grep "^{"
done | tee $O/probe_cfg2.txt | cut -c1-400
