#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_pw_last_gpu.py -q 2>&1 | tail -25 | tee gpurun_out/r3d_last.txt
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_pw_last_gpu.py 2>&1 | tail -8 | tee gpurun_out/r3d_tests.txt
python bench.py --no-roofline --no-cpu-baseline --steps 60 --warmup 10 2>&1 | tail -1 > gpurun_out/r3d_bench.json
python -c "import json; d=json.load(open('gpurun_out/r3d_bench.json')); print('bench', round(d['ms_per_step'],4), d.get('unpipelined',{}).get('ms_per_step'))"
python - <<'PY'
import os
os.environ["X"]="1"
PY
python -c "
import sys; sys.path.insert(0,'.')
from istnet_amd.pointnet2 import fused_mlp
fused_mlp.USE_POOL_EPILOGUE=False
import bench, runpy
sys.argv=['bench.py','--no-roofline','--no-cpu-baseline','--steps','60','--warmup','10']
runpy.run_path('bench.py', run_name='__main__')
" 2>&1 | tail -1 > gpurun_out/r3d_bench_old.json
python -c "import json; d=json.load(open('gpurun_out/r3d_bench_old.json')); print('bench (stored activation)', round(d['ms_per_step'],4), d.get('unpipelined',{}).get('ms_per_step'))"
