#!/bin/bash
# round 3: the measurements that go to profiles/ (run from the repo root on the GPU box)
mkdir -p gpurun_out/r03
O=gpurun_out/r03
bash tools/pmc_traffic.sh r03 > $O/pmc_traffic.txt 2>&1
cp gpurun_out/pmc_r03_traffic.json $O/pmc_traffic.json
rm -rf gpurun_out/pmc_r03_FETCH_SIZE gpurun_out/pmc_r03_WRITE_SIZE
cp $O/pmc_traffic.json profiles/r03_pmc_traffic.json    # bench.py reads it (source hash checked) for roofline.traffic
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_r03 -o enc -- python /root/repo/bench.py --no-roofline --no-cpu-baseline --no-unpipelined --steps 20 --warmup 5 > /root/repo/$O/prof_encoder.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_r03/enc_results.db 29 > $O/encoder_kernel_stats.txt
for pat in pw_fwd pw_dgrad pw_wgrad pw_bwd fps; do python tools/rocprof_summary.py gpurun_out/prof_r03/enc_results.db 29 $pat; done > $O/encoder_gemm_shapes.txt
rm -rf gpurun_out/prof_r03
python bench.py 2>$O/bench_final.err | tail -1 > $O/bench_final.json
python bench.py --workload sa_layer --steps 200 --warmup 20 2>/dev/null | tail -1 > $O/bench_sa_layer.json
python bench.py --workload istnet --no-roofline --steps 30 --warmup 5 2>/dev/null | tail -1 > $O/bench_istnet_full_model.json
python bench.py --workload infer --no-roofline --steps 30 --warmup 5 2>/dev/null | tail -1 > $O/bench_infer_full_model.json
python bench.py --workload istnet --force-dist --no-roofline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_istnet_force_dist.json
python bench.py --no-prefetch --no-roofline --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | tail -1 > $O/bench_noprefetch.json
python tools/aten_in_step.py encoder 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" > $O/framework_kernels_encoder.txt
python tools/aten_in_step.py istnet 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" > $O/framework_kernels_istnet.txt
python tools/bench_fps_chain.py 2>&1 | grep -v amdgpu.ids > $O/fps_chain.txt
python tools/gemm_launch_table.py 2>&1 | grep -v amdgpu.ids > $O/encoder_gemm_launch_table.txt
python tools/step_timeline.py 2>&1 | grep -v amdgpu.ids > $O/step_timeline.txt
python tools/istnet_kernel_times.py --timeline 2>/dev/null > $O/istnet_kernel_times.txt
python tools/istnet_kernel_times.py --infer --timeline 2>/dev/null > $O/infer_kernel_times.txt
ISTNET_SCALE_STREAMS=1 python tools/istnet_step_timeline.py 2>/dev/null > $O/istnet_step_timeline_graph_with_side_streams.txt
ISTNET_SCALE_STREAMS=0 ISTNET_DEFERRED_WGRAD=0 python tools/istnet_step_timeline.py 2>/dev/null > $O/istnet_step_timeline_graph.txt
python tools/aten_sources.py 2>/dev/null > $O/framework_kernels_istnet_by_source_line.txt
python bench.py --workload istnet --no-tuned-gemms --no-roofline --steps 30 --warmup 5 2>/dev/null | tail -1 > $O/bench_istnet_untuned_gemms.json
for m in default tunable; do python tools/exp/decoder_gemm_libs.py $m 2>&1 | grep -v "Warning\|amdgpu.ids"; done > $O/decoder_gemm_libraries.txt
for f in bench_final bench_sa_layer bench_istnet_full_model bench_istnet_untuned_gemms bench_infer_full_model bench_istnet_force_dist bench_noprefetch; do python -c "
import json; d=json.load(open('$O/$f.json')); print('$f', round(d['ms_per_step'],4), round(d['value'],1), (d.get('roofline') or {}).get('frac'), (d.get('unpipelined') or {}).get('ms_per_step'))"; done
head -8 $O/encoder_kernel_stats.txt | cut -c1-150
