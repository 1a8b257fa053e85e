cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ranges.py tests/test_gpu_configs.py tests/test_gpu_mapdev.py -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for w in cfg2 cfg5 cfg4; do
  echo "$w"; timeout 300 python tools/prof_target.py --workload $w --calls 2000 2>> $O/target.err | tee -a $O/targets.txt
done
for w in cfg2 cfg5; do
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$w -o kt -- python tools/prof_target.py --workload $w --calls 300 > /dev/null 2> $O/kt_$w.err
echo "trace $w"; python tools/prof_summary.py $(find $O/kt_$w -name "*.db" | head -1) 2>&1 | grep -i "k_pass"
done
find $O -name "*.db" -delete
