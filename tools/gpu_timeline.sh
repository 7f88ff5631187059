#!/bin/bash
TAG=${1:-x}
mkdir -p gpurun_out
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d /root/repo/gpurun_out/tl_$TAG -o enc -- python /root/repo/bench.py --no-roofline --no-cpu-baseline --steps 10 --warmup 3 > /root/repo/gpurun_out/tl_$TAG.log 2>&1)
python tools/timeline.py gpurun_out/tl_$TAG/enc_results.db ${2:-401} --list > gpurun_out/timeline_$TAG.txt
rm -rf gpurun_out/tl_$TAG
head -12 gpurun_out/timeline_$TAG.txt
