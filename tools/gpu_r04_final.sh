#!/bin/bash
# round 4, final validation: the GPU suite twice (flakiness), smoke(), the bench line, two ranks on one GPU
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04final; mkdir -p $O
for i in 1 2; do ( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_$i.log 2>&1; echo "pytest $i rc=$?" | tee -a $O/pytest_$i.log; done
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/smoke.log
( time timeout 900 python bench.py ) > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" | tee -a $O/bench_n1.err
( time KICP_BENCH_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 10 --warmup 2 --comm p2p --pg-backend gloo --no-cpu-baseline --scans 8 ) > $O/bench_2ranks.json 2> $O/bench_2ranks.err; echo "bench2 rc=$?" | tee -a $O/bench_2ranks.err
( time timeout 900 python bench.py --force-comm --comm rccl --no-cpu-baseline --no-pmc --scans 8 --steps 10 ) > $O/bench_force_rccl.json 2> $O/bench_force_rccl.err; echo "bench rccl rc=$?" | tee -a $O/bench_force_rccl.err
for i in 1 2; do grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" $O/pytest_$i.log | tail -3; done; tail -3 $O/smoke.log; tail -c 300 $O/bench_n1.json; echo; tail -c 300 $O/bench_2ranks.json; echo; tail -c 200 $O/bench_force_rccl.json
