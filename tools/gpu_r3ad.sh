#!/bin/bash
mkdir -p gpurun_out
python tools/istnet_kernel_times.py --infer --timeline > gpurun_out/r3ad_infer_kernels.txt 2>gpurun_out/r3ad_err.log; head -50 gpurun_out/r3ad_infer_kernels.txt | cut -c1-170; sed -n '/^# timeline/,$p' gpurun_out/r3ad_infer_kernels.txt | cut -c1-200 | head -50
