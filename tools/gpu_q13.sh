cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/q13; mkdir -p $O
timeout 300 python tools/gpu_dbg.py cfg2 2>&1 | tee $O/dbg_cfg2.txt
timeout 300 python tools/gpu_dbg.py cfg5 2>&1 | tee $O/dbg_cfg5.txt
