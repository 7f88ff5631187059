#!/bin/bash
mkdir -p gpurun_out
python tools/aten_in_step.py istnet 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" > gpurun_out/r3j_aten_istnet.txt
tail -75 gpurun_out/r3j_aten_istnet.txt | cut -c1-200
