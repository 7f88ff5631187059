#!/bin/bash
run() { python bench.py --workload istnet --no-roofline --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), d['config']['launch'])"; }
run "base (4 hw queues)"
GPU_MAX_HW_QUEUES=8 run "8 hw queues"
GPU_MAX_HW_QUEUES=16 run "16 hw queues"
GPU_MAX_HW_QUEUES=2 run "2 hw queues"
GPU_MAX_HW_QUEUES=8 python bench.py --no-roofline --no-cpu-baseline --steps 50 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('encoder, 8 hw queues', round(d['ms_per_step'],4), round(d['unpipelined']['ms_per_step'],4))"
GPU_MAX_HW_QUEUES=16 python bench.py --no-roofline --no-cpu-baseline --steps 50 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('encoder, 16 hw queues', round(d['ms_per_step'],4), round(d['unpipelined']['ms_per_step'],4))"
python bench.py --no-roofline --no-cpu-baseline --steps 50 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('encoder, default', round(d['ms_per_step'],4), round(d['unpipelined']['ms_per_step'],4))"
