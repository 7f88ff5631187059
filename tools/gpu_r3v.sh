#!/bin/bash
mkdir -p gpurun_out
for m in default rocblas hipblaslt tunable; do timeout 600 python tools/exp/decoder_gemm_libs.py $m 2>&1 | grep -v Warning | tail -6; done
ls gpurun_out/ | grep tunable
