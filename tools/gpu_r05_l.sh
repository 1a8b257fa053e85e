# round 5: mid-size scans of the generic kernel (cfg1: 16 384 points) on several resident kernels side by side - tests, in-process A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05l; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_edge.py tests/test_gpu_small.py tests/test_ingest.py tests/test_facade.py -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log | grep -v "version\|Hostname\|Librccl"
A="timeout 300 python tools/ab_option.py"
( $A --workload cfg1 --batch --calls 1024 --blocks 12 --sets batch_threads=0 base batch_threads=4 batch_threads=6 batch_threads=8
  $A --workload cfg1 --batch --calls 512 --blocks 10 --multi --sets batch_threads=0 base batch_threads=4 batch_threads=6 batch_threads=8 ) 2>&1 | grep "^{" | tee $O/ab_batch_threads_cfg1.txt | cut -c1-700
