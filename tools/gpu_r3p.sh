#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
python bench.py --workload istnet --no-roofline --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r3p_istnet.json
python -c "import json; d=json.load(open('gpurun_out/r3p_istnet.json')); print('istnet', round(d['ms_per_step'],3))"
python bench.py --workload infer --no-roofline --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r3p_infer.json
python -c "import json; d=json.load(open('gpurun_out/r3p_infer.json')); print('infer', round(d['ms_per_step'],3), round(d['value'],1))"
