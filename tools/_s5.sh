mkdir -p gpurun_out/s5
python -m pytest tests -m gpu -x -q --tb=short -p no:warnings 2>&1 | tail -30 > gpurun_out/s5/test_full.txt
tail -3 gpurun_out/s5/test_full.txt
tools/exp/ab_vals.sh ISTNET_EXP_XSTATS "0 1" > gpurun_out/s5/ab_xstats.txt 2>&1
cat gpurun_out/s5/ab_xstats.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/s5/prof -o enc -- python /root/repo/bench.py --no-roofline --no-cpu-baseline --no-unpipelined --no-eager-leg --steps 20 --warmup 5 > /root/repo/gpurun_out/s5/prof.log 2>&1)
python tools/rocprof_summary.py gpurun_out/s5/prof/enc_results.db 29 > gpurun_out/s5/kernel_stats.txt
head -70 gpurun_out/s5/kernel_stats.txt
rm -rf gpurun_out/s5/prof
