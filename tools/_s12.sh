python -m pytest tests/test_autograph_gpu.py tests/test_fused_mlp_gpu.py tests/test_golden_gpu.py tests/test_heads_native_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q -p no:warnings 2>&1 | tail -2
python bench.py --no-roofline --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | tail -1 > gpurun_out/s12_bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/s12_bench.json"))
print(d["ms_per_step"], "unpipelined", (d.get("unpipelined") or {}).get("ms_per_step"), "eager", {k: round(v.get("ms_per_step"), 3) for k, v in (d.get("eager") or {}).items() if isinstance(v, dict)})
PY
