#!/bin/bash
run() { python bench.py --workload istnet --no-roofline --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), d['config']['launch'])"; }
ISTNET_FP_SKIP_STREAM=0 ISTNET_SCALE_STREAMS=0 ISTNET_DEFERRED_WGRAD=0 run "no fp skip, no scale, no deferred"
ISTNET_FP_SKIP_STREAM=0 ISTNET_DEFERRED_WGRAD=0 run "no fp skip, no deferred"
ISTNET_SCALE_STREAMS=0 ISTNET_DEFERRED_WGRAD=0 run "no scale, no deferred"
ISTNET_FP_SKIP_STREAM=0 ISTNET_SCALE_STREAMS=0 ISTNET_DEFERRED_WGRAD=0 ISTNET_GEOMETRY_STREAM=0 run "none of the four"
ISTNET_FP_SKIP_STREAM=0 ISTNET_SCALE_STREAMS=0 ISTNET_DEFERRED_WGRAD=0 ISTNET_EARLY_WORLD=0 run "no 3 + no early world"
ISTNET_FP_SKIP_STREAM=0 ISTNET_SCALE_STREAMS=0 ISTNET_DEFERRED_WGRAD=0 python tools/istnet_step_timeline.py 2>/dev/null | grep -n "rgb\|loss fwd\|estimator bwd\|extractor bwd\|transform bwd\|enhancer bwd\|optimizer\|backward returned" | cut -c1-100
