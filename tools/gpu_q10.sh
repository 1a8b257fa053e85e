cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/q10; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python tools/prof_target.py --workload cfg2 --calls 300 > $O/kt.json 2> $O/kt.err; cat $O/kt.json
python tools/prof_summary.py $(find $O/kt -name "*.db" | head -1) 2>&1 | head -8
KICP_AQL=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt0 -o kt -- python tools/prof_target.py --workload cfg2 --calls 300 > $O/kt0.json 2> $O/kt0.err; cat $O/kt0.json
python tools/prof_summary.py $(find $O/kt0 -name "*.db" | head -1) 2>&1 | head -4
timeout 120 rocprofv3 --pmc VALUBusy -d $O/pmc -o pmc -- python tools/prof_target.py --workload cfg2 --calls 100 > /dev/null 2> $O/pmc.err; python tools/prof_counters_json.py $O/c.json k_pass_gather32 18 $(find $O/pmc -name "*.db") 2>&1 | tail -3
find $O -name "*.db" -delete
