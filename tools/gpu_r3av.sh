#!/bin/bash
run() { python bench.py --workload istnet --no-roofline --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), d['config']['launch'])"; }
ISTNET_SCALE_STREAMS=0 ISTNET_DEFERRED_WGRAD=0 run "no scale, no deferred"
ISTNET_SCALE_STREAMS_BWD=0 ISTNET_DEFERRED_WGRAD=0 run "no scale in bwd, no deferred"
ISTNET_SCALE_STREAMS_BWD=0 ISTNET_DEFERRED_WGRAD=0 ISTNET_FP_SKIP_STREAM=0 run "no scale in bwd, no deferred, no fp skip"
ISTNET_SCALE_STREAMS_BWD=0 run "no scale in bwd"
ISTNET_SCALE_STREAMS_BWD=0 ISTNET_FP_SKIP_STREAM=0 run "no scale in bwd, no fp skip"
run base
