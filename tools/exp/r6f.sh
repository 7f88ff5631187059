#!/bin/bash
# round-6 session f: run-to-run distribution of the captured step (identical command, one box): is there a "fast mode"?
O=gpurun_out/r6f; mkdir -p $O
for i in $(seq 1 14); do
  ISTNET_BENCH_DEBUG_PTRS=1 python bench.py --no-roofline --no-cpu-baseline --no-eager-leg --no-other-clouds --no-unpipelined --steps 50 --warmup 10 --windows 3 2>$O/err_$i.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run $i', round(d['ms_per_step'],4), d.get('windows_ms_per_step'), d.get('debug_ptrs'))"
done > $O/dist.txt 2>&1
cat $O/dist.txt
