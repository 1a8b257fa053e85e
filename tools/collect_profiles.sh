cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --steps 200 --warmup 40 --no-cpu-baseline > $O/bench_under_rocprofv3.json 2> $O/kt.err
python tools/prof_summary.py $(ls $O/kt/*.db | head -1) > $O/kernel_trace_stats.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c -d $O/pmc_$n -o pmc -- python bench.py --steps 40 --warmup 10 --no-cpu-baseline > /dev/null 2> $O/pmc_$n.err
  python tools/prof_counters.py $(ls $O/pmc_$n/*.db | head -1) > $O/pmc_$n.txt 2>&1
done
python tools/prof_traffic.py $(ls $O/pmc_FETCH_SIZE/*.db | head -1) $(ls $O/pmc_WRITE_SIZE/*.db | head -1) k_pass $O/pmc_traffic.json > /dev/null 2>&1
for w in cfg1 cfg4 cfg5; do python bench.py --workload $w --steps 200 --warmup 20 --cpu-seconds 6 > $O/bench_$w.json 2> $O/bench_$w.err; done
python tools/bench_presteps.py > $O/presteps.txt 2>&1
python tools/bench_pipeline.py --frames 40 --dump /tmp/pipe.bin > /dev/null 2>&1 && tests/cpp/facade_test pipeline_timed /tmp/pipe.bin > /tmp/pipe.txt && python tools/bench_pipeline.py --frames 40 --check /tmp/pipe.txt --oracle-frames 40 2>&1 | grep -v "^frame [0-9]* ms" > $O/pipeline.txt
rm -rf $O/kt/*.db $O/pmc_*/  # keep the text summaries only
tail -c 600 $O/bench_n1.json; cat $O/kernel_trace_stats.txt | head -8; cat $O/pmc_traffic.json; tail -3 $O/pipeline.txt
