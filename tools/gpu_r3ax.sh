#!/bin/bash
for f in "" "--no-overlap-allreduce"; do python bench.py --workload istnet --force-dist $f --no-roofline --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('force-dist [$f]', round(d['ms_per_step'],3), d['config']['launch'], d['config']['gradient_exchange']['issued'])"; done
