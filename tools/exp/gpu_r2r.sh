#!/bin/bash
mkdir -p gpurun_out
python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>gpurun_out/r2r.err | tail -1 > gpurun_out/r2r_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r2r_bench.json')); print(d['value'], d['ms_per_step']); print(json.dumps(d['roofline'], indent=1))"
tail -3 gpurun_out/r2r.err
python bench.py --workload sa_layer --steps 50 --warmup 10 2>/dev/null | tail -1 | cut -c1-1500
bash tools/pmc_traffic.sh r02 > gpurun_out/r02_pmc_traffic.txt 2>&1; head -12 gpurun_out/r02_pmc_traffic.txt | cut -c1-160
