#!/bin/bash
python -m pytest tests/test_pw_kernels_gpu.py tests/test_fused_mlp_gpu.py -x -q 2>&1 | tail -3
PW_TUNE=19:0 python tools/bench_pw.py 2>&1 | grep "SA4.*128>256" | cut -c1-30,53-78
python tools/bench_pw.py 2>&1 | grep "SA4.*128>256" | cut -c1-30,53-78
B="python bench.py --no-roofline --no-cpu-baseline --no-unpipelined --steps 50 --warmup 10"
ms() { tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2; do
  echo -n "dgrad kernel  : "; ISTNET_PW_TUNE=19:0 $B 2>&1 | ms
  echo -n "loader / MFMA : "; $B 2>&1 | ms
done
