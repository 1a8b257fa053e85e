cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s; mkdir -p $O
export KICP_WAIT_TIMEOUT_S=20
( time timeout 600 python -m pytest tests/test_gpu_shm.py -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log | grep -v "version\|Hostname\|Librccl"
