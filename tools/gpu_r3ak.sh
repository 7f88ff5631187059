#!/bin/bash
timeout 600 python -m pytest tests/test_pw_kernels_gpu.py tests/test_fused_mlp_gpu.py -q -x 2>&1 | tail -2
bash tools/ab.sh 3
for v in base new; do cp tmp_ab/$v.so ist-net_amd/lib/libistnet_pn2.so; echo "== $v"; python tools/gemm_launch_table.py 2>&1 | grep "fwd_sk" | awk '{s+=$1} END {print "fwd_sk us/step (event-timed):", s}'; done
cp tmp_ab/new.so ist-net_amd/lib/libistnet_pn2.so
