#!/bin/bash
O=gpurun_out/r6s; mkdir -p $O
python -m pytest tests/test_fused_mlp_gpu.py tests/test_golden_gpu.py -m gpu -x -q -W ignore 2>&1 | tail -3 | tee $O/test.txt
bash tools/ab.sh 4 --no-eager-leg --no-other-clouds --windows 3 2>&1 | tee $O/ab.txt
tools/gpu_session.sh r6s prof > $O/session.txt 2>&1
grep -n "pw_gather_add\|bn_relu_pool_cols\|bn_bwd_pooled_finalize" gpurun_out/r6s/kernel_stats_1.txt | cut -c1-140
