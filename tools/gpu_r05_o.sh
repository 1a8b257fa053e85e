# round 5: what bounds ordinary launches of small kernels - the one host thread or the dispatch path?  cfg1 as independent registrations on
# 1 / 2 / 4 / 8 lanes (a handle, an HSA queue and a host thread each; small_resident 0 = every pass an ordinary launch)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05o; mkdir -p $O
( timeout 300 python tools/bench_concurrent.py --workload cfg1 --lanes 1 2 4 8 --count 1024
  timeout 300 python tools/bench_concurrent.py --workload cfg1 --lanes 1 2 4 8 --count 512 --multi ) 2>&1 | grep -v "^\[" | tee $O/concurrent_cfg1.txt | cut -c1-600
