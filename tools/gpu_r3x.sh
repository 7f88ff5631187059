#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python bench.py --workload istnet --no-roofline --tune-gemms --eager --steps 2 --warmup 2 2>&1 | tail -2 | cut -c1-200
ls -la gpurun_out/tunableop_gfx950.csv && wc -l gpurun_out/tunableop_gfx950.csv
cp gpurun_out/tunableop_gfx950.csv ist-net_amd/tuning/tunableop_gfx950.csv
for rep in 1 2; do
python bench.py --workload istnet --no-roofline --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r3x_istnet_tuned.json
python -c "import json; d=json.load(open('gpurun_out/r3x_istnet_tuned.json')); print('tuned istnet', round(d['ms_per_step'],3), d['config']['library_gemms'])"
python bench.py --workload istnet --no-roofline --no-tuned-gemms --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r3x_istnet_untuned.json
python -c "import json; d=json.load(open('gpurun_out/r3x_istnet_untuned.json')); print('untuned istnet', round(d['ms_per_step'],3), d['config']['library_gemms'])"
done
python bench.py --workload infer --no-roofline --steps 20 --warmup 5 2>&1 | tail -1 | cut -c1-200
timeout 600 python -m pytest tests/test_golden_gpu.py -q -x -k "rgb or supervised or config_3 or full" 2>&1 | tail -3
