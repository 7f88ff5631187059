#!/bin/bash
mkdir -p gpurun_out
for tune in "20:64,12:512" "20:64,12:768" "20:64,12:1024" "20:64,12:384" "20:64,12:512" "20:64"; do
  ISTNET_PW_TUNE=$tune python bench.py --no-roofline --no-cpu-baseline --no-unpipelined --steps 80 --warmup 10 2>&1 | tail -1 > gpurun_out/r3n_bench.json
  python -c "import json; d=json.load(open('gpurun_out/r3n_bench.json')); print('tune=[$tune]', round(d['ms_per_step'],4))"
done
