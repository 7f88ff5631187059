#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee gpurun_out/r3c_tests.txt
python tools/aten_in_step.py encoder 2>&1 | tail -80 > gpurun_out/r3c_aten_encoder.txt
for tune in "" "12:128" "12:192"; do
  ISTNET_PW_TUNE=$tune python bench.py --no-roofline --no-cpu-baseline --no-unpipelined --steps 60 --warmup 10 2>&1 | tail -1 > gpurun_out/r3c_bench_$tune.json
  python -c "import json,sys; d=json.load(open('gpurun_out/r3c_bench_$tune.json')); print('tune=[$tune]', round(d['ms_per_step'],4))"
done
