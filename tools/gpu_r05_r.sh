cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05r; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_edge.py tests/test_gpu_small.py tests/test_host.py -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log | grep -v "version\|Hostname\|Librccl"
