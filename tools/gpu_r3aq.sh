#!/bin/bash
run() { python bench.py --workload istnet --no-roofline --steps 20 --warmup 5 $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), d['config']['launch'])"; }
run "base graph" ""
ISTNET_RGB_LAST=1 run "rgb_last graph" ""
ISTNET_RGB_LAST=1 run "rgb_last eager" "--eager"
ISTNET_EARLY_WORLD=0 run "no early world graph" ""
ISTNET_EARLY_WORLD=0 ISTNET_RGB_LAST=1 run "no early world + rgb_last graph" ""
run "base graph" ""
