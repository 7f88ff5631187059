#!/bin/bash
# first GPU cycle of round 2: GPU test suite, default bench line, two-rank same-device dry run
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2a_tests.txt
python bench.py 2>gpurun_out/r2a_bench.err | tail -1 > gpurun_out/r2a_bench.json
python bench.py --gpus 2 --backend gloo --same-device --steps 5 --warmup 2 2>gpurun_out/r2a_bench2.err | tail -1 > gpurun_out/r2a_bench_2ranks.json
tail -3 gpurun_out/r2a_tests.txt; cut -c1-400 gpurun_out/r2a_bench.json; cut -c1-300 gpurun_out/r2a_bench_2ranks.json
