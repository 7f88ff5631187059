#!/bin/bash
# tools/exp/ab_vals.sh VAR "v1 v2 ..." : the captured encoder step under each value of an environment variable, two rounds
VAR=$1; VALS=$2
for rep in 1 2; do
  for v in $VALS; do
    for extra in "" "--no-prefetch"; do
      env $VAR=$v python bench.py --no-roofline --no-cpu-baseline --no-eager-leg --steps 50 --warmup 10 --windows 3 $extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', '$extra', round(d['ms_per_step'],4), d.get('windows_ms_per_step'))"
    done
  done
done
