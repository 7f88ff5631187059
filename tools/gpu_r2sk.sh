#!/bin/bash
python -m pytest tests/test_pw_kernels_gpu.py tests/test_fused_mlp_gpu.py -x -q 2>&1 | tail -4
python tools/bench_pw.py 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_pw_sk2.txt
PW_TUNE=15:0 python tools/bench_pw.py 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_pw_sk1.txt
paste <(cut -c1-52 gpurun_out/bench_pw_sk1.txt) <(cut -c29-52 gpurun_out/bench_pw_sk2.txt) | tail -22
B="python bench.py --no-roofline --no-cpu-baseline --no-unpipelined --steps 50 --warmup 10"
ms() { tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2; do
  echo -n "no sk: "; ISTNET_PW_TUNE=15:0 $B 2>&1 | ms
  echo -n "sk   : "; $B 2>&1 | ms
done
