#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_golden_gpu.py tests/test_fused_mlp_gpu.py -q -x -k "train_mode_golden or supervised_loss or config3 or config5 or badly_centred or point_branch_poses or frozen_world" 2>&1 | tail -30 | tee gpurun_out/r3l_tests.txt
