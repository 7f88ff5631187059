run() { echo -n "$* : "; python bench.py "$@" 2>gpurun_out/s17_err.txt | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'], 4), round(d['value'], 1), d.get('config', {}).get('workload'), (d.get('gradient_exchange') or {}).get('issued'), (d.get('split_precision') or {}).get('ms_per_step'))
except Exception as e:
    print('FAILED', e); print(open('gpurun_out/s17_err.txt').read()[-1500:])"; }
run --force-dist --no-roofline --no-cpu-baseline --no-unpipelined --steps 30 --warmup 5
run --force-dist --capture-allreduce --no-roofline --no-cpu-baseline --no-unpipelined --steps 30 --warmup 5
run --eager --no-roofline --no-cpu-baseline --steps 30 --warmup 5
run --workload pipeline_infer --no-roofline --steps 20 --warmup 5
run --workload istnet --split-precision --no-roofline --steps 20 --warmup 5
run --workload infer --split-precision --no-roofline --steps 20 --warmup 5
run --workload istnet --force-dist --no-roofline --steps 15 --warmup 5
python bench.py --gpus 2 --same-device --backend gloo --no-roofline --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -1 | cut -c1-300
