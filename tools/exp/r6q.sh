#!/bin/bash
# round-6 session q: batched loads in the fused finalize kernels -- parity, A/B against the previous commit's library, kernel times
O=gpurun_out/r6q; mkdir -p $O
python -m pytest tests/test_pw_kernels_gpu.py tests/test_fused_mlp_gpu.py tests/test_golden_gpu.py -m gpu -x -q -W ignore 2>&1 | tail -3 > $O/test.txt
cat $O/test.txt
bash tools/ab.sh 4 --no-eager-leg --no-other-clouds --windows 3 2>&1 | tee $O/ab.txt
tools/gpu_session.sh r6q prof > $O/session.txt 2>&1
grep -n "bn_bwd_dense_finalize\|bn_bwd_pooled_finalize" gpurun_out/r6q/kernel_stats_1.txt | cut -c1-140
