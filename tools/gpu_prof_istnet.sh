#!/bin/bash
TAG=${1:-x}
mkdir -p gpurun_out
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/profi_$TAG -o m -- python /root/repo/bench.py --workload istnet --steps 6 --warmup 2 > /root/repo/gpurun_out/profi_$TAG.log 2>&1)
python tools/rocprof_summary.py gpurun_out/profi_$TAG/m_results.db 12 > gpurun_out/profi_${TAG}_summary.txt
rm -rf gpurun_out/profi_$TAG
head -45 gpurun_out/profi_${TAG}_summary.txt | cut -c1-200
