#!/bin/bash
run() { env "$@" python bench.py --no-roofline --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('BENCH [$*]', round(d['value']), round(d['ms_per_step'],4), round(d['unpipelined']['ms_per_step'],4))"; }
run A=1; run ISTNET_WGRAD_ON_GEOMETRY_STREAM=1; run ISTNET_NO_SCALE_STREAMS=1; run ISTNET_NO_DEFER_WGRAD=1; run ISTNET_WGRAD_STREAM_PER_CHAIN=1; run ISTNET_NO_SCALE_STREAMS=1 ISTNET_WGRAD_ON_GEOMETRY_STREAM=1
