#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_cycle.sh <tag> [pytest-args]
# runs the GPU tests, the bench and a kernel-trace profile; leaves summaries under gpurun_out/
TAG=${1:-x}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python bench.py --no-roofline --no-cpu-baseline --steps 30 --warmup 5 2>&1 | tail -1 | tee gpurun_out/bench_$TAG.json | cut -c1-200
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_$TAG -o enc -- python /root/repo/bench.py --no-roofline --no-cpu-baseline --steps 10 --warmup 3 > /root/repo/gpurun_out/prof_$TAG.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_$TAG/enc_results.db 13 > gpurun_out/prof_${TAG}_summary.txt
for pat in pw_fwd pw_dgrad pw_wgrad; do python tools/rocprof_summary.py gpurun_out/prof_$TAG/enc_results.db 13 $pat; done > gpurun_out/prof_${TAG}_shapes.txt
rm -rf gpurun_out/prof_$TAG
head -${2:-32} gpurun_out/prof_${TAG}_summary.txt | cut -c1-160
