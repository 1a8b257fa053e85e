# round 5: cfg1 (16 384-point scans) - four queues against the kernel resident across the scans (one kernel, three scans in flight)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05k; mkdir -p $O
A="timeout 300 python tools/ab_option.py"
( $A --workload cfg1 --batch --calls 1024 --blocks 12 --sets base batch_queues=0 batch_queues=0,batch_depth=1 batch_queues=0,batch_depth=2
  $A --workload cfg1 --batch --calls 512 --blocks 10 --multi --sets base batch_queues=0 batch_queues=0,batch_depth=1 ) 2>&1 | grep "^{" | tee $O/ab_cfg1_modes.txt | cut -c1-600
