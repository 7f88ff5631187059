#!/bin/bash
mkdir -p gpurun_out
echo "== default (tall dgrad tiles)"; python tools/bench_pw.py 2>&1 | cut -c1-100 | tail -34 > gpurun_out/r2f_pw_tall.txt
echo "== forced 64x64 dgrad"; PW_TUNE=4:2 python tools/bench_pw.py 2>&1 | cut -c1-100 | tail -34 > gpurun_out/r2f_pw_64.txt
paste <(cut -c1-30,58-82 gpurun_out/r2f_pw_tall.txt) <(cut -c58-82 gpurun_out/r2f_pw_64.txt)
python -m pytest tests/test_pw_kernels_gpu.py tests/test_fused_mlp_gpu.py tests/test_golden_gpu.py -m gpu -q -x 2>&1 | tail -4
for t in "" "4:2"; do ISTNET_PW_TUNE=$t python bench.py --no-roofline --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('BENCH tune=$t', d['value'], d['ms_per_step'], d['unpipelined']['ms_per_step'])"; done
