#!/bin/bash
# usage (GPU box, repo root): tools/gpu_prof.sh <tag> [extra bench args]
# bench line (no roofline / cpu legs) + rocprofv3 kernel-trace summary of the graph-replayed step
TAG=${1:-x}; shift
mkdir -p gpurun_out
python bench.py --no-roofline --no-cpu-baseline --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 > gpurun_out/bench_$TAG.json
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_$TAG -o enc -- python /root/repo/bench.py --no-roofline --no-cpu-baseline --no-unpipelined --steps 20 --warmup 5 "$@" > /root/repo/gpurun_out/prof_$TAG.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_$TAG/enc_results.db 29 > gpurun_out/prof_${TAG}_summary.txt
for pat in pw_fwd pw_dgrad pw_wgrad; do python tools/rocprof_summary.py gpurun_out/prof_$TAG/enc_results.db 29 $pat; done > gpurun_out/prof_${TAG}_shapes.txt
for pat in pw_bwd_last pw_dw_last pw_last_prep pw_fwd2 bn_finalize_pool pw_bwd_mid; do python tools/rocprof_summary.py gpurun_out/prof_$TAG/enc_results.db 29 $pat; done >> gpurun_out/prof_${TAG}_shapes.txt
rm -rf gpurun_out/prof_$TAG
python -c "
import json; d=json.load(open('gpurun_out/bench_$TAG.json')); print('BENCH', d['value'], d['ms_per_step'], d.get('unpipelined',{}).get('ms_per_step'))"
head -${LINES_SHOWN:-30} gpurun_out/prof_${TAG}_summary.txt | cut -c1-150
