#!/bin/bash
timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_optim.py -q -x 2>&1 | tail -3
for i in 1 2 3; do python bench.py --no-roofline --no-cpu-baseline --steps 50 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['unpipelined']['ms_per_step'],4))"; done
python bench.py --workload istnet --no-roofline --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('istnet', round(d['ms_per_step'],3))"
