#!/bin/bash
mkdir -p gpurun_out
python tools/step_timeline.py > gpurun_out/r3ac_step_timeline.txt 2>&1; cat gpurun_out/r3ac_step_timeline.txt | tail -45
