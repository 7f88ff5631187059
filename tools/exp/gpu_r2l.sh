#!/bin/bash
run() { env "$@" python bench.py --no-roofline --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('BENCH [$*]', round(d['value']), round(d['ms_per_step'],4), round(d['unpipelined']['ms_per_step'],4))"; }
run A=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=64
python bench.py --eager --no-roofline --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('BENCH [eager]', round(d['value']), round(d['ms_per_step'],4))"
