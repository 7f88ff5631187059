mkdir -p gpurun_out/r5h
python -m pytest tests/test_conv_gpu.py -q -x 2>&1 | tail -3
python bench.py --workload istnet --steps 10 --warmup 5 --no-eager-leg --split-precision 2>gpurun_out/r5h/err.txt | tail -1 > gpurun_out/r5h/istnet_split.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5h/istnet_split.json')); print(d['ms_per_step'], json.dumps(d.get('split_precision'), indent=1))
PY
tail -3 gpurun_out/r5h/err.txt
