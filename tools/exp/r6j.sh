#!/bin/bash
# round-6 session j: compact columns (SA level 1) on / off by input distribution
O=gpurun_out/r6j; mkdir -p $O
run() {
  label=$1; shift
  for extra in "" "--no-prefetch"; do
  env "$@" python bench.py --no-roofline --no-cpu-baseline --no-eager-leg --no-other-clouds --no-unpipelined --steps 50 --warmup 10 --windows 3 $extra $CLOUD 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label $CLOUD $extra', round(d['ms_per_step'],4), d.get('windows_ms_per_step'))"
  done
}
{
for c in dense cube shell; do
  CLOUD="--cloud $c"
  run "compact level 1 (default)" X=1
  run "no compaction" ISTNET_COMPACT_LEVELS=
done
} > $O/compact.txt 2>&1
cat $O/compact.txt
