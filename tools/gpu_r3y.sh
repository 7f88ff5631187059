#!/bin/bash
# A/B on one box: wave issue priority of the dependent-chain kernels (build-time ISTNET_MAIN_PRIO)
L=ist-net_amd/lib/libistnet_pn2.so
B="python bench.py --no-roofline --no-cpu-baseline --steps 50 --warmup 10"
for i in 1 2; do
for v in base prio1 prio2 prio3; do
  cp tmp_ab/$v.so $L
  echo -n "$v encoder: "; $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['unpipelined']['ms_per_step'],4))"
done
done
for v in base prio2 prio3; do
  cp tmp_ab/$v.so $L
  echo -n "$v istnet: "; python bench.py --workload istnet --no-roofline --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"
done
cp tmp_ab/base.so $L
