cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/q6; mkdir -p $O
for acq in 2 1 0; do for rel in 2 1 0; do
echo "== acquire $acq release $rel"; KICP_AQL_ACQ=$acq KICP_AQL_REL=$rel timeout 300 python tools/gpu_blocks.py cfg2 256 2>&1 | grep "aql 1" | tee -a $O/fences.txt
done; done
