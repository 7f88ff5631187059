mkdir -p gpurun_out/r5f
python bench.py --workload pipeline --steps 10 --warmup 5 2>gpurun_out/r5f/pipe.err | tail -1 > gpurun_out/r5f/pipeline.json
python bench.py --workload pipeline_infer --steps 10 --warmup 5 2>gpurun_out/r5f/pipei.err | tail -1 > gpurun_out/r5f/pipeline_infer.json
tail -3 gpurun_out/r5f/pipe.err gpurun_out/r5f/pipei.err | grep -v amdgpu
python - <<'PY'
import json
for f in ('pipeline','pipeline_infer'):
    try:
        d=json.load(open('gpurun_out/r5f/%s.json'%f)); print(f, d['ms_per_step'], d['value'], d['preparation'])
    except Exception as e: print(f, 'FAILED', e)
PY
