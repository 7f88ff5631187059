#!/bin/bash
mkdir -p gpurun_out
for W in 2 3 4; do
  ISTNET_HIPCC_FLAGS="-DISTNET_BWD_SMALL_WAVES=$W" python -c "
import sys; sys.path.insert(0,'.')
import istnet_amd
from istnet_amd import build
build.build(force=True)" > /dev/null 2>&1
  echo "== bwd_small waves=$W"
  python tools/bench_bwd_small.py 2>&1 | tail -6
  python bench.py --no-roofline --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('BENCH', d['value'], d['ms_per_step'], d['unpipelined']['ms_per_step'])"
done
python -c "
import sys; sys.path.insert(0,'.')
import istnet_amd
from istnet_amd import build
build.build(force=True)" > /dev/null 2>&1
python -m pytest tests/test_pw_kernels_gpu.py tests/test_fused_mlp_gpu.py tests/test_golden_gpu.py -m gpu -q -x 2>&1 | tail -4
