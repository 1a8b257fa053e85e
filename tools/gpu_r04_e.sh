#!/bin/bash
# round 4, GPU call E: pull-mode uploads - parity of everything that uploads, pipeline timing, bench line
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04e; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
for m in raw vectors; do
  timeout 300 python tools/bench_pipeline.py --frames 40 --mode $m --dump /tmp/pipe_$m.bin > /tmp/cmd_$m.txt 2>&1
  mode=pipeline_timed; [ $m = raw ] && mode=pipeline_timed_raw
  for pull in 1 0; do
    KICP_PULL_UPLOAD=$pull timeout 300 tests/cpp/facade_test $mode /tmp/pipe_$m.bin > /tmp/pipe_${m}_$pull.txt
    timeout 600 python tools/bench_pipeline.py --frames 40 --mode $m --check /tmp/pipe_${m}_$pull.txt --oracle-frames 3 2>&1 | grep "GPU RegisterFrame" > $O/pipeline_${m}_pull$pull.txt
    KICP_PULL_UPLOAD=$pull KICP_TRACE=1 tests/cpp/facade_test $mode /tmp/pipe_$m.bin 2>&1 >/dev/null | tail -12 > $O/pipeline_calls_${m}_pull$pull.txt
  done
done
( time timeout 900 python bench.py ) > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench rc=$?" | tee -a $O/bench_cfg2.err
tail -3 $O/pytest.log; cat $O/pipeline_*_pull*.txt; grep "ingest\|pre_frame" $O/pipeline_calls_raw_pull1.txt $O/pipeline_calls_raw_pull0.txt $O/pipeline_calls_vectors_pull1.txt $O/pipeline_calls_vectors_pull0.txt
