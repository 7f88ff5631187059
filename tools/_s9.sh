mkdir -p gpurun_out/s9
python -m pytest tests -m gpu -x -q --tb=short -p no:warnings 2>&1 | tail -15 > gpurun_out/s9/test.txt
tail -5 gpurun_out/s9/test.txt
python bench.py --no-roofline --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | tail -1 > gpurun_out/s9/bench.json
python -c "
import json
d = json.load(open('gpurun_out/s9/bench.json'))
print(d['ms_per_step'], d['value'], 'unpipelined', (d.get('unpipelined') or {}).get('ms_per_step'), 'eager', {k: (v.get('ms_per_step') if isinstance(v, dict) else v) for k, v in (d.get('eager') or {}).items() if 'note' not in k})"
