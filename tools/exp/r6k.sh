#!/bin/bash
# round-6 session k: adaptive compact columns -- tests, the three distributions, the default line
O=gpurun_out/r6k; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "compact_columns_follow or other_input_distributions or prefetch or autograph or pipeline" -W ignore 2>&1 | tail -5 > $O/test_focus.txt
for c in dense cube shell; do
  for extra in "" "--no-prefetch"; do
    python bench.py --no-roofline --no-cpu-baseline --no-eager-leg --no-other-clouds --no-unpipelined --steps 50 --warmup 10 --windows 3 --cloud $c $extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('auto --cloud $c $extra', round(d['ms_per_step'],4), d.get('windows_ms_per_step'))"
  done
done > $O/auto.txt 2>&1
python bench.py --no-cpu-baseline --no-eager-leg --no-roofline > $O/bench.json 2> $O/bench.err
python -m pytest tests -m gpu -q -W ignore 2>&1 | tail -4 > $O/test.txt
cat $O/test_focus.txt $O/auto.txt; tail -2 $O/test.txt
python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['unpipelined']['ms_per_step'], json.dumps(d['other_distributions'])[:900])"
