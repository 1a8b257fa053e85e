cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_shm.py -m gpu -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -25 $O/pytest.log
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/bench.json')); print(json.dumps({k:d[k] for k in ('value','ms_per_step','config','cpu_baseline')}, indent=0)[:3000]); print(json.dumps(d['roofline'])[:1500])"
tail -3 $O/bench.err
