cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/q9; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_presteps.py tests/test_golden_pipeline.py tests/test_facade.py -m gpu -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
(timeout 200 python tools/bench_pipeline.py --frames 40 --dump /tmp/pipe.bin > /dev/null 2>&1 && KICP_TRACE=0 timeout 200 tests/cpp/facade_test pipeline_timed /tmp/pipe.bin > /tmp/pipe.txt && timeout 400 python tools/bench_pipeline.py --frames 40 --check /tmp/pipe.txt 2>&1 | grep -v "^frame [0-9]* ms" > $O/pipeline.txt); tail -3 $O/pipeline.txt
KICP_TRACE=1 timeout 200 tests/cpp/facade_test pipeline_timed /tmp/pipe.bin 2>&1 | tail -60 | grep "kicp\]" | tail -24
