cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/q11; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q --timeout 300 -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4
for rep in 1 2; do
KICP_AQL=0 timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HIP', j['value'], j['config']['multi_iteration']['scans_per_s'], j['config']['multi_iteration']['ms_per_iteration'], j['config']['scans_per_s_with_host_input_incl_pcie'])"
timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('AQL', j['value'], j['config']['multi_iteration']['scans_per_s'], j['config']['multi_iteration']['ms_per_iteration'], j['config']['scans_per_s_with_host_input_incl_pcie'])"
done
