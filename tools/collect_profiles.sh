# Copies what `bash tools/gpu.sh <tag> tests smoke bench pipeline pipetrace timeline trace counters benchall ranks2` wrote under
# gpurun_out/<tag>/ into profiles/ under this round's names (the judged, committed copies):  bash tools/collect_profiles.sh <tag> [r06]
T=gpurun_out/$1; R=${2:-r06}; P=profiles
cp $T/gpu_tests.txt $P/${R}_pytest_gpu.txt
grep smoke $T/smoke.log | cut -c1-300 > $P/${R}_smoke.txt
for f in bench_n1 bench_cfg1 bench_cfg4 bench_cfg5 bench_2ranks_1gpu_shm bench_2ranks_1gpu_rccl bench_under_rocprofv3 counters_cfg2 counters_cfg5; do cp $T/$f.json $P/${R}_$f.json; done
for f in kernel_trace_stats kernel_trace_cfg2 kernel_trace_cfg5 pipeline_kernel_trace_raw pipeline_kernel_trace_raw_ahead pipeline_kernel_table_raw pipeline_kernel_table_raw_ahead \
         pipeline_timeline_raw pipeline_timeline_raw_ahead pipeline_calls_raw pipeline_calls_raw_ahead; do cp $T/$f.txt $P/${R}_$f.txt; done
for m in raw raw_ahead; do cat $T/pipeline_${m}_1.txt $T/pipeline_${m}_2.txt $T/pipeline_${m}_3.txt > $P/${R}_pipeline_$m.txt; done
[ -f $T/pipeline_ab.txt ] && cp $T/pipeline_ab.txt $P/${R}_pipeline_ab.txt
ls $P | grep "^${R}_" | wc -l
