cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/q14; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ranges.py tests/test_gpu_configs.py -m gpu -q --timeout 600 -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
timeout 300 python tools/gpu_dbg.py cfg2 2>&1 | grep "dbg [013]:" | tee $O/dbg_cfg2.txt
timeout 300 python tools/gpu_dbg.py cfg5 2>&1 | grep "dbg [013]:" | tee $O/dbg_cfg5.txt
