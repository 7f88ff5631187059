#!/bin/bash
for m in "" "--eager"; do for i in 1 2; do python bench.py --workload istnet --no-roofline --steps 20 --warmup 5 $m 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('istnet [$m]', round(d['ms_per_step'],3), d['config']['launch'])"; done; done
