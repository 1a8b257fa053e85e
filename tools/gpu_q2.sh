cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/q2; mkdir -p $O
python tools/bench_pipeline.py --frames 40 --dump /tmp/pipe.bin > /dev/null 2>&1
timeout 200 tests/cpp/facade_test pipeline_steps /tmp/pipe.bin > $O/steps.txt 2>&1; tail -6 $O/steps.txt
timeout 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; tail -c 3000 $O/bench_n1.json
nproc; lscpu | grep "Model name"
