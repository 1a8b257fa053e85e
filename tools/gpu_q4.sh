cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/q4; mkdir -p $O
timeout 300 python tools/gpu_blocks.py cfg2 64,128,256,512 2>&1 | tee $O/blocks_cfg2.txt
timeout 300 python tools/gpu_blocks.py cfg5 256,512 2>&1 | tee $O/blocks_cfg5.txt
