#!/bin/bash
O=gpurun_out/r6g; mkdir -p $O
for i in 1 2 3; do python tools/exp/capture_modes.py 8 --new-streams 2>/dev/null; echo ---; done > $O/capture_modes.txt 2>&1
cat $O/capture_modes.txt
