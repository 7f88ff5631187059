#!/bin/bash
# Kernel timeline of the last captured encoder step (un-pipelined form: one step per replay).  usage: tools/gpu_timeline.sh <tag> [dispatches]
TAG=${1:-x}
mkdir -p gpurun_out
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d /root/repo/gpurun_out/tl_$TAG -o enc -- python /root/repo/bench.py --no-prefetch --no-roofline --no-cpu-baseline --no-eager-leg --windows 1 --steps 10 --warmup 3 > /root/repo/gpurun_out/tl_$TAG.log 2>&1)
python tools/timeline.py gpurun_out/tl_$TAG/enc_results.db ${2:-226} --list > gpurun_out/timeline_$TAG.txt
rm -rf gpurun_out/tl_$TAG
head -12 gpurun_out/timeline_$TAG.txt
