O=gpurun_out/r5k; mkdir -p $O
python bench.py 2>$O/bench_final.err | tail -1 > $O/bench_final.json
python bench.py --steps 20 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_driver_form.json
python bench.py --no-prefetch --no-cpu-baseline --no-roofline --no-eager-leg 2>/dev/null | tail -1 > $O/bench_noprefetch.json
python bench.py --workload sa_layer 2>/dev/null | tail -1 > $O/bench_sa_layer.json
python bench.py --workload istnet --steps 20 --warmup 5 --split-precision 2>/dev/null | tail -1 > $O/bench_istnet_full_model.json
python bench.py --workload infer --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_infer_full_model.json
python bench.py --workload pipeline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_pipeline.json
python bench.py --workload pipeline_infer --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_pipeline_infer.json
python bench.py --force-dist --capture-allreduce --no-cpu-baseline --no-roofline --no-eager-leg 2>/dev/null | tail -1 > $O/bench_encoder_force_dist_captured.json
python bench.py --cpu-threads 128 --steps 5 --warmup 2 --no-roofline --no-eager-leg --no-unpipelined 2>/dev/null | tail -1 > $O/cpu_baseline_all_cores.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5k/*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], round(d['ms_per_step'],3), round(d['value'],1), d['unit'])
    except Exception as e: print(f, 'FAILED', e)
d=json.load(open('gpurun_out/r5k/bench_final.json'))
print({k:round(v['ms_per_step'],3) for k,v in d['eager'].items() if isinstance(v,dict)}, d['eager']['ratio_to_graph_replay'])
r=d['roofline']; print(r['kernel'], r['frac'], r['traffic'], r['traffic_source'], r['all_gemm_kernels_frac'], r['step_flops_frac'], r['dominant_family']['kernel'], r['dominant_family']['frac'])
print(d['unpipelined']['ms_per_step'], d['cpu_baseline']['value'], d['cpu_baseline']['thread_sweep_ms_per_step'])
d=json.load(open('gpurun_out/r5k/cpu_baseline_all_cores.json')); print('128 threads:', d['cpu_baseline']['thread_sweep_ms_per_step'], d['cpu_baseline']['value'])
d=json.load(open('gpurun_out/r5k/bench_istnet_full_model.json')); print('istnet eager', {k:round(v['ms_per_step'],2) for k,v in d.get('eager',{}).items() if isinstance(v,dict)}, d.get('split_precision',{}).get('ms_per_step'))
PY
