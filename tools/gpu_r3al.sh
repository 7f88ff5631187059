#!/bin/bash
L=ist-net_amd/lib/libistnet_pn2.so
for v in base nostore; do cp tmp_ab/$v.so $L; echo "== $v"; python tools/gemm_launch_table.py 2>&1 | grep "fwd_sk" | awk '{s+=$1; n+=$2} END {print "fwd_sk us/step (event-timed):", s, "launches", n}'; done
cp tmp_ab/base.so $L
