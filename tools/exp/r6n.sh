#!/bin/bash
O=gpurun_out/r6n; mkdir -p $O
ab() {
  label=$1; shift
  for extra in "" "--no-prefetch"; do
    env "$@" python bench.py --no-roofline --no-cpu-baseline --no-eager-leg --no-other-clouds --no-unpipelined --steps 50 --warmup 10 --windows 3 $extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', '$extra', round(d['ms_per_step'],4), d.get('windows_ms_per_step'))"
  done
}
for rep in 1 2 3; do
  ab "fps 1 wave (default)" X=1
  ab "fps 4 waves         " ISTNET_PN2_TUNE=0:1024
  ab "fps 2 waves         " ISTNET_PN2_TUNE=0:1024,3:2
done > $O/ab.txt 2>&1
cat $O/ab.txt
