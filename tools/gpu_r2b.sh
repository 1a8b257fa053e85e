cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ranges.py tests/test_gpu_configs.py tests/test_gpu_mapdev.py -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for opt in "" "--option lanes_per_query=2" "--option block=256" "--option block=64"; do
  echo "cfg2 $opt"; timeout 200 python tools/prof_target.py --workload cfg2 --calls 3000 $opt 2>> $O/target.err | tee -a $O/targets.txt
done
echo "cfg2 events"; timeout 200 python tools/prof_target.py --workload cfg2 --calls 2000 --events 2>> $O/target.err | tee -a $O/targets.txt
echo "cfg2 G2 events"; timeout 200 python tools/prof_target.py --workload cfg2 --calls 2000 --events --option lanes_per_query=2 2>> $O/target.err | tee -a $O/targets.txt
echo "cfg5 events"; timeout 300 python tools/prof_target.py --workload cfg5 --calls 500 --events 2>> $O/target.err | tee -a $O/targets.txt
echo "cfg5 b256"; timeout 300 python tools/prof_target.py --workload cfg5 --calls 500 --events --option block=256 2>> $O/target.err | tee -a $O/targets.txt
echo "cfg4"; timeout 300 python tools/prof_target.py --workload cfg4 --calls 3000 --events 2>> $O/target.err | tee -a $O/targets.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_cfg2 -o kt -- python tools/prof_target.py --workload cfg2 --calls 300 > $O/kt_cfg2.json 2> $O/kt_cfg2.err
python tools/prof_summary.py $(find $O/kt_cfg2 -name "*.db" | head -1) 2>&1 | head -4
find $O -name "*.db" -delete
