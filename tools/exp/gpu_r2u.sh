#!/bin/bash
python -m pytest tests/test_fused_mlp_gpu.py -m gpu -q -x -k "concat_free" 2>&1 | tail -25
python -m pytest tests/test_golden_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x 2>&1 | tail -4
for w in istnet; do python bench.py --workload $w --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('BENCH $w', round(d['value'],1), round(d['ms_per_step'],3))"; done
