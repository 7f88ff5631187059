#!/bin/bash
python tools/exp/corun.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_pw_kernels_gpu.py -x -q -k "bwd_mid" 2>&1 | tail -2
python tools/bench_bwd_mid.py 2>&1 | tail -11
B="python bench.py --no-roofline --no-cpu-baseline --no-unpipelined --steps 50 --warmup 10"
ms() { tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2 3; do echo -n "step: "; $B 2>&1 | ms; done
