#!/bin/bash
for rep in 1 2; do
for e in "ISTNET_COMPACT_LEVELS=0" "ISTNET_COMPACT_LEVELS=0,1" "ISTNET_COMPACT_LEVELS=0,1,2" "ISTNET_NO_COMPACT=1"; do env $e python bench.py --no-roofline --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('BENCH [$e]', round(d['value']), round(d['ms_per_step'],4), round(d['unpipelined']['ms_per_step'],4))"; done
done
