#!/bin/bash
# A/B of the working tree against HEAD's fused_mlp.py on one box (python-side change only)
B="python bench.py --no-roofline --no-cpu-baseline --no-unpipelined --steps 50 --warmup 10"
ms() { tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
cp ist-net_amd/pointnet2/fused_mlp.py /tmp/new_fused.py
for i in 1 2 3; do
  cp tmp_ab/head_fused_mlp.py ist-net_amd/pointnet2/fused_mlp.py; echo -n "HEAD: "; $B 2>&1 | ms
  cp /tmp/new_fused.py ist-net_amd/pointnet2/fused_mlp.py; echo -n "new : "; $B 2>&1 | ms
done
