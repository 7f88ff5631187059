mkdir -p gpurun_out/s7
python -m pytest tests/test_ops_gpu.py tests/test_golden_gpu.py tests/test_fused_mlp_gpu.py tests/test_pipeline_gpu.py tests/test_autograph_gpu.py -m gpu -x -q --tb=short -p no:warnings 2>&1 | tail -15 > gpurun_out/s7/test.txt
tail -5 gpurun_out/s7/test.txt
tools/exp/ab_vals.sh ISTNET_EXP_BQPAIR "0 1" > gpurun_out/s7/ab_bqpair.txt 2>&1
cat gpurun_out/s7/ab_bqpair.txt
for v in 0 1; do ISTNET_EXP_BQPAIR=$v python bench.py --no-roofline --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('BQPAIR=$v', d['ms_per_step'], 'unpipelined', (d.get('unpipelined') or {}).get('ms_per_step'), 'eager', {k: (v.get('ms_per_step') if isinstance(v, dict) else v) for k, v in (d.get('eager') or {}).items()})"; done | tee gpurun_out/s7/eager.txt
