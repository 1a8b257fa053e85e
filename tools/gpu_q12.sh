cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/q12; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_mapdev.py tests/test_golden_pipeline.py tests/test_facade.py -m gpu -q --timeout 300 2>&1 | tail -3
KICP_TRACE=0 timeout 300 python tools/bench_mapupdate.py 2>&1 | tee $O/mapupdate.txt
timeout 120 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python tools/bench_mapupdate.py > /dev/null 2>&1; python tools/prof_summary.py $(find $O/kt -name "*.db" | head -1) 2>&1 | grep -v "250112\|gather32" | head -14; find $O -name "*.db" -delete
