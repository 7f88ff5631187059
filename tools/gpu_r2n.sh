#!/bin/bash
run() { env "$@" python bench.py --no-roofline --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('BENCH [$*]', round(d['value']), round(d['ms_per_step'],4), round(d['unpipelined']['ms_per_step'],4))"; }
run A=1; run ISTNET_NO_WGRAD_PARK=1; run A=1; run ISTNET_NO_WGRAD_PARK=1
python -m pytest tests/test_fused_mlp_gpu.py tests/test_golden_gpu.py tests/test_pipeline_gpu.py tests/test_optim.py -m gpu -q -x 2>&1 | tail -3
