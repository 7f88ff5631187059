mkdir -p gpurun_out/r5g
python -m pytest tests/test_preprocess.py tests/test_capi.py -q 2>&1 | tail -4
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r5g/prof -o fill -- python /root/repo/tools/prof_fill.py > /root/repo/gpurun_out/r5g/prof.log 2>&1)
python tools/rocprof_summary.py gpurun_out/r5g/prof/fill_results.db 12 > gpurun_out/r5g/fill_kernel_stats_new.txt
rm -rf gpurun_out/r5g/prof
head -16 gpurun_out/r5g/fill_kernel_stats_new.txt | cut -c1-150
bash tools/_s2.sh
