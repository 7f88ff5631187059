#!/bin/bash
# round 3, first cycle: full GPU suite, FPS chain microbench, bench (pipelined + unpipelined)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r3a_tests.txt
python tools/bench_fps_chain.py 2>&1 | tee gpurun_out/r3a_fps_chain.txt
python bench.py --no-roofline --no-cpu-baseline --steps 40 --warmup 8 2>&1 | tail -1 | tee gpurun_out/r3a_bench.json | cut -c1-400
