#!/bin/bash
run() { python bench.py --workload istnet --no-roofline --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), d['config']['launch'])"; }
run "default (side streams off)"
ISTNET_DEFERRED_WGRAD_HEADS=1 run "heads wgrad deferred"
ISTNET_DEFERRED_WGRAD_HEADS=1 run "heads wgrad deferred"
run "default (side streams off)"
