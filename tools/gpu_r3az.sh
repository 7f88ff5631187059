#!/bin/bash
run() { python bench.py --no-roofline --no-cpu-baseline --steps 50 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), round(d['unpipelined']['ms_per_step'],4))"; }
run base
ISTNET_DEFERRED_WGRAD=0 run "no deferred"
ISTNET_FP_SKIP_STREAM=0 run "no fp skip"
ISTNET_SCALE_STREAMS_BWD=0 run "no scale in bwd"
ISTNET_SCALE_STREAMS=0 run "no scale"
ISTNET_GEOMETRY_STREAM=0 run "no geometry stream"
run base
