mkdir -p gpurun_out/pipe
timeout 900 python -m pytest tests/test_gpu_edge.py tests/test_gpu_small.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/pipe/tests13.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/pipe/tests13.txt | tail -5
for m in "" "--multi"; do timeout 300 python tools/ab_option.py --workload cfg4 --sets batch_queues=0 batch_queues=3 batch_queues=4 batch_queues=6 --batch $m --calls 512 --blocks 20; done 2>&1 | grep "^{"
timeout 300 python tools/ab_option.py --workload cfg2 --sets batch_queues=0 batch_queues=4 --batch --calls 512 --blocks 16 2>&1 | grep "^{"
