mkdir -p gpurun_out/pipe; rm -f gpurun_out/pipe/ab4.txt
timeout 900 python -m pytest tests/test_gpu_edge.py tests/test_gpu_small.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3 > gpurun_out/pipe/tests4.txt
for w in cfg2 cfg1 cfg4; do
  timeout 300 python tools/ab_option.py --workload $w --option batch_depth --values 1 2 3 4 --batch --calls 256 --blocks 24 >> gpurun_out/pipe/ab4.txt 2>&1
  timeout 300 python tools/ab_option.py --workload $w --option batch_depth --values 1 2 3 4 --batch --calls 256 --blocks 24 --multi >> gpurun_out/pipe/ab4.txt 2>&1
done
cat gpurun_out/pipe/tests4.txt; grep workload gpurun_out/pipe/ab4.txt
