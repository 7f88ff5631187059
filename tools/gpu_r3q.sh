#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_rgb_ops_gpu.py tests/test_golden_gpu.py -q -x 2>&1 | tail -12
python bench.py --workload istnet --no-roofline --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r3q_istnet.json
python -c "import json; d=json.load(open('gpurun_out/r3q_istnet.json')); print('istnet', round(d['ms_per_step'],3))"
python tools/bench_rgb.py 2>&1 | tail -3
