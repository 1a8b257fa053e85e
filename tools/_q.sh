mkdir -p gpurun_out/pipe
timeout 900 python -m pytest tests/test_gpu_edge.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/pipe/tests15.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/pipe/tests15.txt | tail -3
for w in cfg2 cfg1; do for m in "" "--multi"; do timeout 300 python tools/ab_option.py --workload $w --option dbg --values 12 0 --batch $m --calls 256 --blocks 16; done; done 2>&1 | grep "^{"
timeout 300 python tools/ab_option.py --workload cfg2 --option dbg --values 12 0 --fixed batch_queues=0 --batch --calls 256 --blocks 16 2>&1 | grep "^{"
