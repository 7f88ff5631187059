#!/bin/bash
# A/B of one environment switch on the captured encoder step: tools/exp/ab_env.sh VAR [bench args...]; alternates 0 / 1 three times
VAR=$1; shift
for rep in 1 2 3; do
  for v in 0 1; do
    for extra in "" "--no-prefetch"; do
      env $VAR=$v python bench.py --no-roofline --no-cpu-baseline --no-eager-leg --steps 50 --warmup 10 --windows 3 $extra "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', '$extra', round(d['ms_per_step'],4), d.get('windows_ms_per_step'))"
    done
  done
done
