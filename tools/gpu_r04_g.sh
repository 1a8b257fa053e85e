#!/bin/bash
# round 4, GPU call G: scan fused into compaction / gather - parity of the pre-steps, pipeline timing
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04g; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_presteps.py tests/test_facade.py tests/test_golden_pipeline.py tests/test_ingest.py tests/test_gpu_mapdev.py -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
for m in raw vectors; do
  timeout 300 python tools/bench_pipeline.py --frames 40 --mode $m --dump /tmp/pipe_$m.bin > /tmp/cmd_$m.txt 2>&1
  mode=pipeline_timed; [ $m = raw ] && mode=pipeline_timed_raw
  timeout 300 tests/cpp/facade_test $mode /tmp/pipe_$m.bin > /tmp/pipe_$m.txt
  timeout 600 python tools/bench_pipeline.py --frames 40 --mode $m --check /tmp/pipe_$m.txt --oracle-frames 3 2>&1 | grep "GPU RegisterFrame" > $O/pipeline_$m.txt
  KICP_TRACE=1 tests/cpp/facade_test $mode /tmp/pipe_$m.bin 2>&1 >/dev/null | tail -12 > $O/pipeline_calls_$m.txt
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_pipe -o kt -- tests/cpp/facade_test pipeline_timed_raw /tmp/pipe_raw.bin > /dev/null 2> $O/kt_pipe.err; python tools/prof_summary.py $(find $O/kt_pipe -name "*.db" | head -1) > $O/pipeline_kernel_trace.txt 2>&1; rm -rf $O/kt_pipe
tail -3 $O/pytest.log; cat $O/pipeline_raw.txt $O/pipeline_vectors.txt; cat $O/pipeline_calls_raw.txt; head -30 $O/pipeline_kernel_trace.txt
