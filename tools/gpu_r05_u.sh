cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05u; mkdir -p $O
for i in 1 2 3; do ( time timeout 600 python -m pytest tests/test_gpu_shm.py tests/test_gpu_multirank.py -x -q ) > $O/pytest_$i.log 2>&1; echo "pytest $i rc=$?"; tail -4 $O/pytest_$i.log | grep -v "version\|Hostname\|Librccl"; done
