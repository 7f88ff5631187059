#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pw_kernels_gpu.py -q -x 2>&1 | tail -2
bash tools/ab.sh 3
