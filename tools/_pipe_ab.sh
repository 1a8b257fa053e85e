mkdir -p gpurun_out/pipe; rm -f gpurun_out/pipe/ab8.txt
timeout 900 python -m pytest tests/test_gpu_edge.py tests/test_gpu_small.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/pipe/tests8.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/pipe/tests8.txt | tail -3
for w in cfg2 cfg1; do
  timeout 300 python tools/ab_option.py --workload $w --sets batch_depth=1 batch_depth=2 batch_depth=3 --batch --calls 256 --blocks 24 >> gpurun_out/pipe/ab8.txt 2>&1
done
grep workload gpurun_out/pipe/ab8.txt
python tools/trace_batch.py cfg2 --depths 3 --pass 40
timeout 300 python bench.py > gpurun_out/pipe/bench8.json 2> gpurun_out/pipe/bench8.err; head -c 600 gpurun_out/pipe/bench8.json
