#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04j; mkdir -p $O
timeout 300 python tools/bench_pipeline.py --frames 12 --mode raw --dump /tmp/pipe_raw.bin > /dev/null 2>&1
for pre in ""; do
  rm -rf $O/kt
  LD_PRELOAD=$pre KICP_ROCTX=1 timeout 300 rocprofv3 --marker-trace --kernel-trace -d $O/kt -o kt -- tests/cpp/facade_test pipeline_timed_raw /tmp/pipe_raw.bin > /dev/null 2> $O/err.txt
  db=$(find $O/kt -name "*.db" | head -1)
  echo "preload=[$pre] db=$db"
  python tools/prof_markers.py "$db" | tee $O/pipeline_roctx_ranges.txt | head -16
done
rm -rf $O/kt
