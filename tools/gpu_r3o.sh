#!/bin/bash
mkdir -p gpurun_out
for tune in "" "8:96" "8:160" "8:256" "5:384" "5:768" "14:768" "14:1536" "16:768" "16:1536" "7:256" "7:512" "1:384,2:384" "1:768,2:768" ""; do
  ISTNET_PW_TUNE=$tune python bench.py --no-roofline --no-cpu-baseline --no-unpipelined --steps 80 --warmup 10 2>&1 | tail -1 > gpurun_out/r3o_bench.json
  python -c "import json; d=json.load(open('gpurun_out/r3o_bench.json')); print('tune=[$tune]', round(d['ms_per_step'],4))"
done
