#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_pw_last_gpu.py -q 2>&1 | tail -5
LINES_SHOWN=12 bash tools/gpu_prof.sh r3f_new
grep -B1 -A8 "pw_bwd_last'" gpurun_out/prof_r3f_new_shapes.txt | head -12
grep -A8 "matching 'pw_dw_last'" gpurun_out/prof_r3f_new_shapes.txt | head -10
grep -A4 "matching 'pw_last_prep'" gpurun_out/prof_r3f_new_shapes.txt
grep -A11 "matching 'pw_fwd2'" gpurun_out/prof_r3f_new_shapes.txt
grep -A4 "matching 'bn_finalize_pool'" gpurun_out/prof_r3f_new_shapes.txt
