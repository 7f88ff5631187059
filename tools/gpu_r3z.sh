#!/bin/bash
mkdir -p gpurun_out
python tools/gemm_launch_table.py > gpurun_out/r3z_gemm_launches.txt 2>&1; head -90 gpurun_out/r3z_gemm_launches.txt
