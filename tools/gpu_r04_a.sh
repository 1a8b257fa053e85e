#!/bin/bash
# round 4, GPU call A: the whole GPU suite, the two-rank soak, the reworked bench line
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04a; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
( time timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/soak_two_ranks.py --cycles 200 --device 0 ) > $O/soak.log 2>&1; echo "soak rc=$?" | tee -a $O/soak.log
( time timeout 900 python bench.py ) > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench rc=$?" | tee -a $O/bench_cfg2.err
( time KICP_BENCH_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 10 --warmup 2 --comm shm --pg-backend gloo --no-cpu-baseline --scans 8 ) > $O/bench_2ranks.json 2> $O/bench_2ranks.err; echo "bench2 rc=$?" | tee -a $O/bench_2ranks.err
tail -3 $O/pytest.log; tail -2 $O/soak.log; tail -c 600 $O/bench_cfg2.json; tail -c 400 $O/bench_2ranks.json
