# round 5: the pre-steps' look-ahead thread after its error-path tidy-up (tests), and scans in flight on more queues for scans that leave
# the machine mostly empty (cfg1: 256 waves per scan)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05j; mkdir -p $O
( time timeout 900 python -m pytest tests/test_ingest.py tests/test_facade.py tests/test_gpu_presteps.py tests/test_golden_pipeline.py -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log | grep -v "version\|Hostname\|Librccl"
A="timeout 300 python tools/ab_option.py"
( $A --workload cfg1 --batch --calls 1024 --blocks 12 --sets base batch_queues=6 batch_queues=8
  $A --workload cfg1 --batch --calls 512 --blocks 10 --multi --sets base batch_queues=6 batch_queues=8 ) 2>&1 | grep "^{" | tee $O/ab_queues_small.txt | cut -c1-500
