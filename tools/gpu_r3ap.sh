#!/bin/bash
mkdir -p gpurun_out
python tools/istnet_kernel_times.py --graph --timeline 2>/dev/null > gpurun_out/r3ap_graph.txt; head -30 gpurun_out/r3ap_graph.txt | cut -c1-150; sed -n '/^# timeline/,$p' gpurun_out/r3ap_graph.txt | cut -c1-170 | head -80
