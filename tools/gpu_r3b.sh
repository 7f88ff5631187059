#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee gpurun_out/r3b_tests.txt
python bench.py --no-roofline --no-cpu-baseline --steps 40 --warmup 8 2>&1 | tail -1 | tee gpurun_out/r3b_bench.json | cut -c1-300
python bench.py --no-roofline --no-cpu-baseline --steps 40 --warmup 8 2>&1 | tail -1 | tee gpurun_out/r3b_bench2.json | cut -c1-300
