#!/bin/bash
# round 4, GPU call B: batch-resident kernel - parity test, A/B against the plain loop
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04b; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_edge.py tests/test_gpu_small.py tests/test_gpu_parity.py -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
for w in cfg2 cfg1; do
  timeout 300 python tools/ab_option.py --workload $w --batch --calls 256 --blocks 30 --option batch_resident --values 0 1 > $O/ab_batch_$w.json 2> $O/ab_batch_$w.err
  timeout 300 python tools/ab_option.py --workload $w --batch --multi --calls 128 --blocks 30 --option batch_resident --values 0 1 > $O/ab_batch_multi_$w.json 2>> $O/ab_batch_$w.err
done
tail -3 $O/pytest.log; cat $O/ab_batch_*.json
