#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pw_kernels_gpu.py tests/test_fused_mlp_gpu.py -q -x 2>&1 | tail -2
bash tools/ab.sh 3
python tools/gemm_launch_table.py 2>&1 | grep "fwd_sk\|timed GEMM"
