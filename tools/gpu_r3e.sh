#!/bin/bash
mkdir -p gpurun_out
LINES_SHOWN=45 bash tools/gpu_prof.sh r3e_new
grep -A100 "pw_bwd_last" gpurun_out/prof_r3e_new_shapes.txt | head -60
