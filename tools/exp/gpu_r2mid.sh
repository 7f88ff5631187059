#!/bin/bash
# A/B of the fused mid-size-layer backward inside the training step (one box)
cd "$(dirname "$0")/../.." && B="python bench.py --no-roofline --no-cpu-baseline --steps 50 --warmup 10"
ms() { tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2; do
  echo -n "pair : "; ISTNET_PW_TUNE=9:0 $B 2>&1 | ms
  echo -n "fused: "; $B 2>&1 | ms
  echo -n "fused 256: "; ISTNET_PW_TUNE=8:256 $B 2>&1 | ms
done
