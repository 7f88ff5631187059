#!/bin/bash
mkdir -p gpurun_out
python tools/istnet_kernel_times.py > gpurun_out/r3t_istnet_kernels.txt 2>gpurun_out/r3t_err.log; head -100 gpurun_out/r3t_istnet_kernels.txt | cut -c1-190; tail -3 gpurun_out/r3t_err.log
