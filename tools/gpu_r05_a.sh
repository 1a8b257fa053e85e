# round 5, first GPU call: the new tie tests + the whole GPU suite, VALU attribution of the four-waves build (this build and round 4's
# side by side on the same box), a bench line, the batch rate of both builds alternating
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_ties.py -x -q ) > $O/pytest_ties.log 2>&1; echo "ties rc=$?"; tail -15 $O/pytest_ties.log
( time timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_ties.py ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
bash tools/valu_attribution.sh $O/new
bash tools/valu_attribution.sh $O/r04 $GRAFT_REPO_ROOT/tools/ab/libkicp_amd_r04.so
for rep in 1 2; do
  for lib in "" $GRAFT_REPO_ROOT/tools/ab/libkicp_amd_r04.so; do
    KICP_AB_LIB=$lib timeout 200 python tools/prof_target.py --workload cfg2 --calls 8192 --batch 256 2>&1 | tail -1 | sed "s|^|lib=${lib:-new} |" | tee -a $O/ab_batch_rate.txt
    KICP_AB_LIB=$lib timeout 200 python tools/prof_target.py --workload cfg2 --calls 2048 --batch 256 --multi 2>&1 | tail -1 | sed "s|^|lib=${lib:-new} multi |" | tee -a $O/ab_batch_rate.txt
  done
done
timeout 500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; tail -c 600 $O/bench_n1.json; echo; tail -3 $O/bench_n1.err
du -sh $O
