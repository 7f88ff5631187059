#!/bin/bash
mkdir -p gpurun_out
python tools/istnet_kernel_times.py 2>/dev/null > gpurun_out/r3ay_eager.txt; head -7 gpurun_out/r3ay_eager.txt | cut -c1-120; grep -n "igemm\|ck::\|_ZN2ck\|Cijk\|naive\|SubTensor\|gemm" gpurun_out/r3ay_eager.txt | cut -c1-140 | head -50
