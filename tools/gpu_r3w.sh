#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_golden_gpu.py tests/test_pipeline_gpu.py -q -x 2>&1 | tail -3
for v in 1 0; do
  ISTNET_EARLY_WORLD=$v python bench.py --workload istnet --no-roofline --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r3w_istnet_$v.json
  python -c "import json; d=json.load(open('gpurun_out/r3w_istnet_$v.json')); print('early_world=$v istnet', round(d['ms_per_step'],3))"
done
timeout 1200 python bench.py --workload istnet --no-roofline --tune-gemms --eager --steps 2 --warmup 2 2>&1 | tail -2 | cut -c1-200
ls -la gpurun_out/tunableop_gfx950.csv && wc -l gpurun_out/tunableop_gfx950.csv
cp gpurun_out/tunableop_gfx950.csv ist-net_amd/tuning/tunableop_gfx950.csv
python bench.py --workload istnet --no-roofline --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r3w_istnet_tuned.json
python -c "import json; d=json.load(open('gpurun_out/r3w_istnet_tuned.json')); print('tuned istnet', round(d['ms_per_step'],3), d['config']['library_gemms'])"
python bench.py --workload infer --no-roofline --steps 20 --warmup 5 2>&1 | tail -1 | cut -c1-200
