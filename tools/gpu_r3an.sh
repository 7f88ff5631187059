#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_optim.py tests/test_pipeline_gpu.py tests/test_rgb_ops_gpu.py tests/test_golden_gpu.py -q -x 2>&1 | tail -3
for i in 1 2; do python bench.py --workload istnet --no-roofline --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('istnet', round(d['ms_per_step'],3))"; done
python bench.py --workload istnet --force-dist --no-roofline --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('istnet force-dist', round(d['ms_per_step'],3))"
python tools/aten_sources.py 2>/dev/null > gpurun_out/r3an_aten_sources.txt; head -14 gpurun_out/r3an_aten_sources.txt | cut -c1-170
