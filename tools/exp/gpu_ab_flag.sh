#!/bin/bash
# usage: tools/exp/gpu_ab_flag.sh FLAG [rounds]
F=$1; R=${2:-3}
B="--no-roofline --no-cpu-baseline --steps 50 --warmup 10"
for i in $(seq $R); do
  for v in 0 1; do echo -n "$F=$v: "; python tools/exp/ab_flag.py $F=$v $B 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; done
done
