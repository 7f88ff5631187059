#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_heads_native_gpu.py tests/test_ops_gpu.py -q -x 2>&1 | tail -3
python tools/istnet_kernel_times.py --timeline > gpurun_out/r3u_istnet_kernels.txt 2>gpurun_out/r3u_err.log; head -40 gpurun_out/r3u_istnet_kernels.txt | cut -c1-170; tail -3 gpurun_out/r3u_err.log
python bench.py --workload istnet --no-roofline --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r3u_istnet.json
python -c "import json; d=json.load(open('gpurun_out/r3u_istnet.json')); print('istnet', round(d['ms_per_step'],3))"
python bench.py --no-roofline --steps 30 --warmup 5 2>&1 | tail -1 | cut -c1-300
