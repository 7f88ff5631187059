mkdir -p gpurun_out/r5j
bash tools/pmc_traffic.sh r05 > gpurun_out/r5j/pmc_table.txt 2>&1
cp gpurun_out/pmc_r05_traffic.json gpurun_out/r5j/ 2>/dev/null
head -12 gpurun_out/r5j/pmc_table.txt | cut -c1-160
rm -rf gpurun_out/pmc_r05_FETCH_SIZE gpurun_out/pmc_r05_WRITE_SIZE
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r5j/prof -o enc -- python /root/repo/bench.py --no-roofline --no-cpu-baseline --no-unpipelined --no-eager-leg --steps 20 --warmup 5 --windows 1 > /root/repo/gpurun_out/r5j/prof.log 2>&1)
python tools/rocprof_summary.py gpurun_out/r5j/prof/enc_results.db 29 > gpurun_out/r5j/encoder_kernel_stats.txt
rm -rf gpurun_out/r5j/prof
head -8 gpurun_out/r5j/encoder_kernel_stats.txt | cut -c1-150
