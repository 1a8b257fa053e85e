cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for w in cfg2 cfg5; do
for opt in "--option xcds=8" "--option xcds=1" "--option xcds=8 --option block=128"; do
  echo "$w $opt"; timeout 300 python tools/prof_target.py --workload $w --calls 2000 $opt 2>> $O/target.err | tee -a $O/targets.txt
done; done
for w in cfg2 cfg5; do for x in 8 1; do
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt${x}_$w -o kt -- python tools/prof_target.py --workload $w --calls 300 --option xcds=$x > /dev/null 2> $O/kt${x}_$w.err
echo "trace $w xcds=$x"; python tools/prof_summary.py $(find $O/kt${x}_$w -name "*.db" | head -1) 2>&1 | grep -i "k_pass"
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/pmc${x}_$w -o pmc -- python tools/prof_target.py --workload $w --calls 300 --option xcds=$x > /dev/null 2> $O/pmc${x}_$w.err
python tools/prof_counters.py $(find $O/pmc${x}_$w -name "*.db" | head -1) 2>&1 | grep -i "k_pass"
done; done
find $O -name "*.db" -delete
