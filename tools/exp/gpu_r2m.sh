#!/bin/bash
for q in 5 6 8; do echo "== queues $q"; DEBUG_HIP_FORCE_GRAPH_QUEUES=$q python bench.py --no-roofline --no-cpu-baseline --steps 40 --warmup 8 2>&1 | tail -4 | cut -c1-300; done
