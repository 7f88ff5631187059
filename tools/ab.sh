#!/bin/bash
# A/B on one GPU box: ab_base/base.so (built from git HEAD by tools/build_base.sh) vs the working-tree library.
# usage: tools/ab.sh [rounds] [bench args...]
R=${1:-2}; shift
B="python bench.py --no-roofline --no-cpu-baseline --steps 50 --warmup 10 $*"
L=ist-net_amd/lib/libistnet_pn2.so
cp $L ab_base/new.so
for i in $(seq $R); do
  cp ab_base/base.so $L; echo -n "base: "; $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], (d.get('unpipelined') or {}).get('ms_per_step', ''))"
  cp ab_base/new.so $L;  echo -n "new:  "; $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], (d.get('unpipelined') or {}).get('ms_per_step', ''))"
done
