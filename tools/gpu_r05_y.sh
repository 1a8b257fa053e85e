# round 5, at the round's last code commit: the drop-in pipeline on 131 072-point PointCloud2 frames, raw and with the next message announced
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05y; mkdir -p $O
timeout 300 python tools/bench_pipeline.py --frames 40 --mode raw --dump /tmp/pipe.bin > /dev/null 2>&1
for m in raw raw_ahead; do
  mode=pipeline_timed_raw; [ $m = raw_ahead ] && mode=pipeline_timed_raw_ahead
  for rep in 1 2; do
    timeout 300 tests/cpp/facade_test $mode /tmp/pipe.bin > /tmp/pipe_$m.txt
    timeout 600 python tools/bench_pipeline.py --frames 40 --mode $m --check /tmp/pipe_$m.txt --oracle-frames 0 --ref-frames 0 2>&1 | grep -v "^frame [0-9]* ms" > $O/pipeline_${m}_$rep.txt; grep "GPU RegisterFrame" $O/pipeline_${m}_$rep.txt | cut -c1-220
  done
done
