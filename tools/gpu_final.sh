#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_final.sh <tag>
# the default bench line + a rocprofv3 kernel-trace summary of the same step (pipelined HIP-graph replay only)
TAG=${1:-x}
mkdir -p gpurun_out
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_$TAG.json
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_$TAG -o enc -- python /root/repo/bench.py --no-roofline --no-cpu-baseline --no-unpipelined --steps 20 --warmup 5 > /root/repo/gpurun_out/prof_$TAG.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_$TAG/enc_results.db > gpurun_out/prof_${TAG}_summary.txt
for pat in pw_fwd pw_dgrad pw_wgrad; do python tools/rocprof_summary.py gpurun_out/prof_$TAG/enc_results.db 29 $pat; done > gpurun_out/prof_${TAG}_shapes.txt
rm -rf gpurun_out/prof_$TAG
python - <<PY
import json
d = json.load(open("gpurun_out/bench_$TAG.json"))
r = d["roofline"]
print(d["value"], d["ms_per_step"], r["kernel"], r["launches_per_step"], r["avg_launch_us"], r["achieved"], r["frac"])
PY
head -8 gpurun_out/prof_${TAG}_summary.txt | cut -c1-150
