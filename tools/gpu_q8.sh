cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/q8; mkdir -p $O
for rep in 1 2; do timeout 300 python tools/gpu_blocks.py cfg2 256 2>&1 | grep "dbg 0\|dbg 8" | tee -a $O/ab.txt; done
for rep in 1 2; do
KICP_AQL=0 timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HIP', j['value'], j['config']['multi_iteration']['scans_per_s'], j['config']['scans_per_s_with_host_input_incl_pcie'])"
timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('AQL', j['value'], j['config']['multi_iteration']['scans_per_s'], j['config']['scans_per_s_with_host_input_incl_pcie'])"
done
