#!/bin/bash
mkdir -p gpurun_out
python bench.py --workload istnet --no-roofline --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r3i_istnet.json
python -c "import json; d=json.load(open('gpurun_out/r3i_istnet.json')); print('istnet', round(d['ms_per_step'],3))"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_r3i -o ist -- python /root/repo/bench.py --workload istnet --no-roofline --steps 10 --warmup 3 > /root/repo/gpurun_out/prof_r3i.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_r3i/ist_results.db 17 > gpurun_out/r3i_istnet_kernels.txt
head -5 gpurun_out/r3i_istnet_kernels.txt
rm -rf gpurun_out/prof_r3i
