#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_ist -o ist -- python /root/repo/bench.py --workload istnet --no-roofline --steps 6 --warmup 3 > /root/repo/gpurun_out/prof_ist.log 2>&1
cd /root/repo
ls -la gpurun_out/prof_ist/ 2>/dev/null | head; find gpurun_out/prof_ist -name "*.csv" | head
f=$(find gpurun_out/prof_ist -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && head -40 "$f" | cut -c1-160
tail -3 gpurun_out/prof_ist.log | cut -c1-200
