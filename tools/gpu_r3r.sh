#!/bin/bash
# fused BatchNorm + ReLU of the RGB trunk: parity, full-model step, branch breakdown; config-1 line
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_rgb_ops_gpu.py tests/test_golden_gpu.py -q -x 2>&1 | tail -12
python bench.py --workload istnet --no-roofline --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r3r_istnet.json
python -c "import json; d=json.load(open('gpurun_out/r3r_istnet.json')); print('istnet', round(d['ms_per_step'],3))"
python tools/bench_rgb.py 2>&1 | tail -3
python tools/profile_rgb.py > gpurun_out/r3r_rgb_breakdown.txt 2>&1; head -45 gpurun_out/r3r_rgb_breakdown.txt | cut -c1-150
python bench.py --workload sa_layer 2>&1 | tail -1 > gpurun_out/r3r_sa_layer.json; cut -c1-400 gpurun_out/r3r_sa_layer.json
