#!/bin/bash
mkdir -p gpurun_out
python tools/istnet_step_timeline.py 2>gpurun_out/r3ar_err.log | tee gpurun_out/r3ar_timeline_graph.txt | head -60; tail -3 gpurun_out/r3ar_err.log
python tools/istnet_step_timeline.py --eager 2>>gpurun_out/r3ar_err.log | tee gpurun_out/r3ar_timeline_eager.txt | head -60
