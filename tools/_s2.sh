mkdir -p gpurun_out/r5e
python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/r5e/gputests.txt 2>&1
tail -12 gpurun_out/r5e/gputests.txt
python bench.py --steps 20 --warmup 10 --no-cpu-baseline 2>gpurun_out/r5e/bench.err | tail -1 > gpurun_out/r5e/bench.json
python bench.py --workload istnet --steps 10 --warmup 5 --no-eager-leg 2>/dev/null | tail -1 > gpurun_out/r5e/istnet.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5e/bench.json'))
print(d['ms_per_step'], d['windows_ms_per_step'], d['unpipelined']['ms_per_step'], {k:v['ms_per_step'] for k,v in d['eager'].items() if isinstance(v,dict)})
d=json.load(open('gpurun_out/r5e/istnet.json')); print('istnet', d['ms_per_step'], d.get('windows_ms_per_step'))
PY
