#!/bin/bash
# round 4, GPU call F: batch-resident wave kernel (parity, cfg4 A/B), far-voxel flag test, bench lines of cfg4 / cfg1
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04f; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_edge.py tests/test_gpu_small.py tests/test_gpu_ranges.py -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
timeout 300 python tools/ab_option.py --workload cfg4 --batch --calls 256 --blocks 30 --option batch_resident --values 0 1 > $O/ab_batch_cfg4.json 2> $O/ab.err
timeout 300 python tools/ab_option.py --workload cfg4 --batch --multi --calls 128 --blocks 30 --option batch_resident --values 0 1 > $O/ab_batch_multi_cfg4.json 2>> $O/ab.err
( time timeout 600 python bench.py --workload cfg4 --no-pmc ) > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "bench rc=$?" | tee -a $O/bench_cfg4.err
tail -3 $O/pytest.log; cat $O/ab_batch_cfg4.json $O/ab_batch_multi_cfg4.json; tail -c 1500 $O/bench_cfg4.json | head -c 600
