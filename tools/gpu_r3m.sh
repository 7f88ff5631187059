#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
for i in 1 2; do
python bench.py --no-roofline --no-cpu-baseline --steps 60 --warmup 10 2>&1 | tail -1 > gpurun_out/r3m_bench_$i.json
python -c "import json; d=json.load(open('gpurun_out/r3m_bench_$i.json')); print('bench', round(d['ms_per_step'],4), d.get('unpipelined',{}).get('ms_per_step'))"
done
