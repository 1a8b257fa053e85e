#!/bin/bash
# quick check of the pieces the collection run depends on: roctx marker trace, ingest alignment, fused scans
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04h; mkdir -p $O
( timeout 600 python -m pytest tests/test_ingest.py tests/test_gpu_presteps.py tests/test_facade.py -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
timeout 300 python tools/bench_pipeline.py --frames 40 --mode raw --dump /tmp/pipe_raw.bin > /dev/null 2>&1
for i in 1 2 3 4 5; do timeout 300 tests/cpp/facade_test pipeline_timed_raw /tmp/pipe_raw.bin > /tmp/pipe_raw.txt; timeout 600 python tools/bench_pipeline.py --frames 40 --mode raw --check /tmp/pipe_raw.txt --oracle-frames 0 2>&1 | grep "GPU RegisterFrame"; done | tee $O/pipeline_raw_3x.txt
KICP_TRACE=1 tests/cpp/facade_test pipeline_timed_raw /tmp/pipe_raw.bin 2>&1 >/dev/null | tail -12 > $O/pipeline_calls_raw.txt; cat $O/pipeline_calls_raw.txt
KICP_ROCTX=1 timeout 300 rocprofv3 --marker-trace --kernel-trace -d $O/kt_roctx -o kt -- tests/cpp/facade_test pipeline_timed_raw /tmp/pipe_raw.bin > /dev/null 2> $O/kt_roctx.err
python tools/prof_markers.py $(find $O/kt_roctx -name "*.db" | head -1) > $O/pipeline_roctx_ranges.txt 2>&1; cat $O/pipeline_roctx_ranges.txt | head -20
python tools/prof_summary.py $(find $O/kt_roctx -name "*.db" | head -1) 2>&1 | grep "k_ingest\|k_compact\|k_scan" 
rm -rf $O/kt_roctx
