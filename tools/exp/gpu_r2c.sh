#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_pw_kernels_gpu.py tests/test_pipeline_gpu.py tests/test_golden_gpu.py tests/test_ops_gpu.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r2c_tests.txt
python tools/bench_csr.py > gpurun_out/r2c_bench_csr.txt 2>&1
python bench.py --no-roofline --no-cpu-baseline --steps 30 --warmup 5 2>&1 | tail -1 > gpurun_out/r2c_bench.json
tail -4 gpurun_out/r2c_tests.txt; cat gpurun_out/r2c_bench_csr.txt; cut -c1-330 gpurun_out/r2c_bench.json
