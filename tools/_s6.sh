mkdir -p gpurun_out/s6
for mode in "" "--no-prefetch"; do
tag=pipe; [ -n "$mode" ] && tag=nopre
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/s6/prof_$tag -o enc -- python /root/repo/bench.py --no-roofline --no-cpu-baseline --no-unpipelined --no-eager-leg --steps 20 --warmup 5 --windows 1 $mode > /root/repo/gpurun_out/s6/prof_$tag.log 2>&1)
python tools/rocprof_summary.py gpurun_out/s6/prof_$tag/enc_results.db > gpurun_out/s6/kernel_stats_$tag.txt
python tools/step_kernel_list.py gpurun_out/s6/prof_$tag/enc_results.db 0 > gpurun_out/s6/timeline_$tag.txt
rm -rf gpurun_out/s6/prof_$tag
done
head -4 gpurun_out/s6/kernel_stats_pipe.txt; head -3 gpurun_out/s6/timeline_pipe.txt
