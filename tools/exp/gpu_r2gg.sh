#!/bin/bash
# group_points_grad over inverse lists vs the LDS-atomic kernel; index-op microbench under rocprofv3
python -m pytest tests/test_ops_gpu.py tests/test_golden_gpu.py -x -q -m gpu 2>&1 | tail -3
python tools/bench_idx.py 2>&1 | grep -E "group_grad|interp_grad"
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_idx -- python $R/tools/bench_idx.py > /dev/null 2>&1
cd $R; python tools/rocprof_summary.py $(ls gpurun_out/prof_idx/*/*_results.db | head -1) | head -24
