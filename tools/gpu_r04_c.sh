#!/bin/bash
# round 4, GPU call C: chained pre-steps (parity + pipeline timing), resident-pass timeline
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04c; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_presteps.py tests/test_facade.py tests/test_golden_pipeline.py tests/test_ingest.py tests/test_gpu_edge.py -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
for m in raw vectors; do
  timeout 300 python tools/bench_pipeline.py --frames 40 --mode $m --dump /tmp/pipe_$m.bin > /tmp/cmd_$m.txt 2>&1
  mode=pipeline_timed; [ $m = raw ] && mode=pipeline_timed_raw
  timeout 300 tests/cpp/facade_test $mode /tmp/pipe_$m.bin > /tmp/pipe_$m.txt
  timeout 600 python tools/bench_pipeline.py --frames 40 --mode $m --check /tmp/pipe_$m.txt --oracle-frames 40 2>&1 | grep -v "^frame [0-9]* ms" > $O/pipeline_$m.txt
  KICP_TRACE=1 tests/cpp/facade_test $mode /tmp/pipe_$m.bin 2>&1 >/dev/null | tail -12 > $O/pipeline_calls_$m.txt
done
timeout 300 python tools/trace_resident.py cfg2 > $O/trace_resident_cfg2.txt 2>&1
timeout 300 python tools/trace_resident.py cfg1 > $O/trace_resident_cfg1.txt 2>&1
tail -3 $O/pytest.log; head -4 $O/pipeline_raw.txt; tail -3 $O/pipeline_raw.txt; head -4 $O/pipeline_vectors.txt; cat $O/pipeline_calls_raw.txt; cat $O/trace_resident_cfg2.txt
