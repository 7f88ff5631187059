#!/bin/bash
# A/B on one box: fused BatchNorm + ReLU of the RGB trunk on / off
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_rgb_ops_gpu.py -q -x 2>&1 | tail -3
for rep in 1 2; do
for v in 1 0; do
  ISTNET_FUSED_TRUNK_NORM=$v python bench.py --workload istnet --no-roofline --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r3s_istnet_$v.json
  python -c "import json; d=json.load(open('gpurun_out/r3s_istnet_$v.json')); print('fused_trunk=$v istnet', round(d['ms_per_step'],3))"
done
done
