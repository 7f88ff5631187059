#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_fused_mlp_gpu.py -m gpu -q -x -k "compact" 2>&1 | tail -40
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -12
for e in "" "ISTNET_NO_COMPACT=1"; do env $e python bench.py --no-roofline --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('BENCH [$e]', d['value'], d['ms_per_step'], d['unpipelined']['ms_per_step'])"; done
