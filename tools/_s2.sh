mkdir -p gpurun_out/r5c
python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/r5c/gputests.txt 2>&1
tail -15 gpurun_out/r5c/gputests.txt
python bench.py --steps 20 --warmup 10 2>gpurun_out/r5c/bench.err | tail -1 > gpurun_out/r5c/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5c/bench.json'))
print(d['ms_per_step'], d['windows_ms_per_step'], d['unpipelined']['ms_per_step'])
print(json.dumps(d['eager'], indent=1))
r=d['roofline']; print({k:r[k] for k in r if k in ('frac','all_gemm_kernels_frac','step_flops_frac','dominant_family','all_gemm_kernels_two_roof_model_frac','traffic','traffic_source')})
print(d['cpu_baseline'])
PY
