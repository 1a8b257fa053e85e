# round 5, at the round's last code commit: the kernel trace of the bench command (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05x; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_bench -o kt -- python bench.py --steps 10 --no-cpu-baseline --no-pmc --scans 16 > $O/bench_under_rocprofv3.json 2> $O/kt_bench.err; echo "rc=$?"
python tools/prof_summary.py $(find $O/kt_bench -name "*.db" | head -1) > $O/kernel_trace_stats.txt 2>&1; head -12 $O/kernel_trace_stats.txt | cut -c1-200
rm -rf $O/kt_bench
