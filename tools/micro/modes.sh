# Where the caller of the drop-in runs on a two-socket host: the look-ahead drive (tests/cpp/facade_test) six times each - unbound, bound to
# eight CPUs of node 0, to every other CPU, to 32 CPUs - with the host's topology printed first.  The figures of INTEGRATION.md section 5
# ("where the host side runs") come from this script:  gpurun -- 'bash tools/micro/modes.sh'
cd $GRAFT_REPO_ROOT
lscpu | grep -i "model name\|socket\|core(s)\|thread(s)\|numa\|^CPU(s)" 
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; taskset -p $$
timeout 300 python tools/bench_pipeline.py --frames 40 --mode raw --dump /tmp/pipe.bin > /dev/null 2>&1
run() { for i in 1 2 3 4 5 6; do "$@" tests/cpp/facade_test pipeline_timed_raw_ahead /tmp/pipe.bin 2>/dev/null | grep -i "drive\|frames in" | head -1 | cut -c1-90; done; }
echo "== default"; run env
echo "== taskset 0-7"; run taskset -c 0-7
echo "== taskset 0,2,4,6,8,10,12,14"; run taskset -c 0,2,4,6,8,10,12,14
echo "== numactl-ish: taskset 0-31"; run taskset -c 0-31
