#!/bin/bash
run() { python bench.py --workload istnet --no-roofline --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), d['config']['launch'])"; }
run "world estimator first"
ISTNET_WORLD_ESTIMATOR_FIRST=0 run "world estimator last (reference order)"
run "world estimator first"
ISTNET_WORLD_ESTIMATOR_FIRST=0 run "world estimator last (reference order)"
timeout 600 python -m pytest tests/test_golden_gpu.py -q -x 2>&1 | tail -2
