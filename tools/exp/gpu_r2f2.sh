#!/bin/bash
python -m pytest tests/test_pw_kernels_gpu.py tests/test_fused_mlp_gpu.py -x -q 2>&1 | tail -4
python tools/bench_pw.py 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_pw_f2.txt
PW_TUNE=13:0 python tools/bench_pw.py 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_pw_f1.txt
paste <(cut -c1-52 gpurun_out/bench_pw_f1.txt) <(cut -c29-52 gpurun_out/bench_pw_f2.txt)
B="python bench.py --no-roofline --no-cpu-baseline --no-unpipelined --steps 50 --warmup 10"
ms() { tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2; do
  echo -n "fwd v1: "; ISTNET_PW_TUNE=13:0 $B 2>&1 | ms
  echo -n "fwd v2: "; $B 2>&1 | ms
done
