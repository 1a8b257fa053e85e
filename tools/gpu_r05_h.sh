# round 5: the chained pre-steps with the look-ahead upload and the frame's download queued by threads of their own - tests, timing, traces
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05h; mkdir -p $O
( time timeout 900 python -m pytest tests/test_ingest.py tests/test_facade.py tests/test_gpu_presteps.py tests/test_golden_pipeline.py -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log | grep -v "version\|Hostname\|Librccl"
timeout 300 python tools/bench_pipeline.py --frames 40 --mode raw --dump /tmp/pipe.bin > /dev/null 2>&1
for m in raw raw_ahead; do
  mode=pipeline_timed_raw; [ $m = raw_ahead ] && mode=pipeline_timed_raw_ahead
  for rep in 1 2 3; do
    timeout 300 tests/cpp/facade_test $mode /tmp/pipe.bin > /tmp/pipe_$m.txt
    timeout 900 python tools/bench_pipeline.py --frames 40 --mode $m --check /tmp/pipe_$m.txt --oracle-frames 0 --ref-frames 0 2>&1 | grep -v "^frame [0-9]* ms" > $O/pipeline_${m}_$rep.txt; grep "GPU RegisterFrame" $O/pipeline_${m}_$rep.txt | cut -c1-220
    python - /tmp/pipe_$m.txt <<'PY'
import sys, numpy as np
ms = np.array([float(l.split()[3]) for l in open(sys.argv[1]) if l.startswith("frame")])[5:]
print("  frames 5..: " + " ".join("%.3f" % x for x in np.percentile(ms, [5, 25, 50, 75, 95])) + "  (p5 p25 p50 p75 p95)")
PY
  done
  KICP_TRACE=1 tests/cpp/facade_test $mode /tmp/pipe.bin 2>&1 >/dev/null | tail -78 | head -16 > $O/pipeline_calls_$m.txt
  echo "== $m"; cat $O/pipeline_calls_$m.txt
done
