cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05v; mkdir -p $O
timeout 300 python bench.py --workload cfg1 --cpu-seconds 2 --scans 16 --no-pmc --no-cpu-baseline > $O/bench_cfg1.json 2> $O/bench_cfg1.err; echo "cfg1 rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05v/bench_cfg1.json")); print(d["value"], d["config"]["scans_in_flight"])
PY
python -c "
import os
print('affinity of a fresh python:', len(os.sched_getaffinity(0)))
import torch
print('after import torch:', len(os.sched_getaffinity(0)))
"
