#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_heads_native_gpu.py -q 2>&1 | tail -25 | tee gpurun_out/r3k_heads.txt
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_heads_native_gpu.py 2>&1 | tail -6 | tee gpurun_out/r3k_tests.txt
python bench.py --workload istnet --no-roofline --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r3k_istnet.json
python -c "import json; d=json.load(open('gpurun_out/r3k_istnet.json')); print('istnet', round(d['ms_per_step'],3))"
