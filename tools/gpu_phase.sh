#!/bin/bash
L=ist-net_amd/lib/libistnet_pn2.so
cp $L tmp_ab/prod.so
for v in phase_v4 phase_v5; do
  [ -f tmp_ab/$v.so ] || continue
  cp tmp_ab/$v.so $L
  echo "== $v"
  python tools/fwd_sk_phases.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/fwd_sk_$v.txt | head -10
done
cp tmp_ab/prod.so $L
