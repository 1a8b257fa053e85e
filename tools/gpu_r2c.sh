cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ranges.py tests/test_gpu_configs.py -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for w in cfg2 cfg5; do
for opt in "--option compact=1 --option block=256" "--option compact=1 --option block=128" "--option compact=0 --option block=256" "--option compact=0 --option block=128"; do
  echo "$w $opt"; timeout 300 python tools/prof_target.py --workload $w --calls 2000 $opt 2>> $O/target.err | tee -a $O/targets.txt
done; done
echo "cfg2 multi"; timeout 300 python tools/prof_target.py --workload cfg2 --calls 1000 --multi 2>> $O/target.err | tee -a $O/targets.txt
echo "cfg4"; timeout 300 python tools/prof_target.py --workload cfg4 --calls 3000 2>> $O/target.err | tee -a $O/targets.txt
for w in cfg2 cfg5; do
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$w -o kt -- python tools/prof_target.py --workload $w --calls 300 > $O/kt_$w.json 2> $O/kt_$w.err
python tools/prof_summary.py $(find $O/kt_$w -name "*.db" | head -1) 2>&1 | grep -i "k_pass"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt0_$w -o kt -- python tools/prof_target.py --workload $w --calls 300 --option compact=0 > $O/kt0_$w.json 2> $O/kt0_$w.err
python tools/prof_summary.py $(find $O/kt0_$w -name "*.db" | head -1) 2>&1 | grep -i "k_pass"
done
find $O -name "*.db" -delete
