#!/bin/bash
O=gpurun_out/r6m; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -m gpu -x -q -W ignore -k "fps" 2>&1 | tail -4 > $O/test.txt
python tools/exp/bench_fps_waves.py > $O/fps_waves.txt 2>&1
cat $O/test.txt; grep -v amdgpu $O/fps_waves.txt
