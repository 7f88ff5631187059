#!/bin/bash
# round 4: the measurements that go to profiles/ (run from the repo root on the GPU box)
O=gpurun_out/r04; mkdir -p $O
bash tools/pmc_traffic.sh r04 > $O/pmc_traffic.txt 2>&1
cp gpurun_out/pmc_r04_traffic.json $O/pmc_traffic.json; cp $O/pmc_traffic.json profiles/r04_pmc_traffic.json    # bench.py reads it (source hash checked)
rm -rf gpurun_out/pmc_r04_FETCH_SIZE gpurun_out/pmc_r04_WRITE_SIZE
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_r04 -o enc -- python /root/repo/bench.py --no-roofline --no-cpu-baseline --no-unpipelined --steps 20 --warmup 5 > /root/repo/$O/prof_encoder.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_r04/enc_results.db 29 > $O/encoder_kernel_stats.txt
for pat in pw_fwd pw_dgrad pw_wgrad pw_bwd fps; do python tools/rocprof_summary.py gpurun_out/prof_r04/enc_results.db 29 $pat; done > $O/encoder_gemm_shapes.txt
rm -rf gpurun_out/prof_r04
python bench.py 2>$O/bench_final.err | tail -1 > $O/bench_final.json
python bench.py --workload sa_layer --steps 200 --warmup 20 2>/dev/null | tail -1 > $O/bench_sa_layer.json
python bench.py --workload istnet --no-roofline --steps 30 --warmup 5 2>/dev/null | tail -1 > $O/bench_istnet_full_model.json
python bench.py --workload infer --no-roofline --steps 30 --warmup 5 2>/dev/null | tail -1 > $O/bench_infer_full_model.json
python bench.py --workload istnet --force-dist --no-roofline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_istnet_force_dist.json
python bench.py --workload istnet --force-dist --capture-allreduce --no-roofline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_istnet_force_dist_captured.json
python bench.py --force-dist --capture-allreduce --no-roofline --no-cpu-baseline --no-unpipelined --steps 30 --warmup 5 2>/dev/null | tail -1 > $O/bench_encoder_force_dist_captured.json
python bench.py --force-dist --no-roofline --no-cpu-baseline --no-unpipelined --steps 30 --warmup 5 2>/dev/null | tail -1 > $O/bench_encoder_force_dist.json
python bench.py --no-prefetch --no-roofline --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | tail -1 > $O/bench_noprefetch.json
python tools/aten_in_step.py encoder 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" > $O/framework_kernels_encoder.txt
python tools/aten_in_step.py istnet 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" > $O/framework_kernels_istnet.txt
python tools/aten_sources.py 2>/dev/null > $O/framework_kernels_istnet_by_source_line.txt
python tools/gemm_launch_table.py 2>&1 | grep -v amdgpu.ids > $O/encoder_gemm_launch_table.txt
python tools/step_timeline.py 2>&1 | grep -v amdgpu.ids > $O/step_timeline.txt
ISTNET_SCALE_STREAMS=0 ISTNET_DEFERRED_WGRAD=0 python tools/istnet_step_timeline.py 2>/dev/null > $O/istnet_step_timeline_graph.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d /root/repo/$O/prof_ist -o ist -- python /root/repo/bench.py --workload istnet --no-roofline --steps 20 --warmup 5 > /root/repo/$O/prof_ist.log 2>&1)
grep '^{"metric' $O/prof_ist.log | tail -1 > $O/bench_istnet_traced.json
MS2=$(python -c "import json; print(json.load(open('$O/bench_istnet_traced.json'))['ms_per_step'])")
python tools/rocprof_libsplit.py $O/prof_ist/ist_results.db 20 $MS2 1.0 > $O/istnet_kernel_times.txt 2>&1
rm -rf $O/prof_ist $O/prof_ist.log
cp ist-net_amd/lib/libistnet_pn2.so /tmp/prod.so; cp ab_base/phase.so ist-net_amd/lib/libistnet_pn2.so
python tools/bwd_mid_phases.py 2>&1 | grep -v amdgpu.ids > $O/bwd_mid_phases.txt
cp /tmp/prod.so ist-net_amd/lib/libistnet_pn2.so
python tools/bench_interp_grad.py 2>&1 | grep -v amdgpu.ids > $O/interp_grad_microbench.txt
python tools/bench_scatter.py 2>&1 | grep -v amdgpu.ids > $O/scatter_microbench.txt
python tools/exp/capture_nested_fork.py 2>&1 | grep -v amdgpu.ids > $O/capture_nested_fork.txt
python tools/exp/capture_fork_autograd.py 2>&1 | grep -v amdgpu.ids > $O/capture_fork_autograd.txt
for f in bench_final bench_sa_layer bench_istnet_full_model bench_infer_full_model bench_istnet_force_dist bench_istnet_force_dist_captured bench_encoder_force_dist bench_encoder_force_dist_captured bench_noprefetch; do python -c "
import json; d=json.load(open('$O/$f.json')); print('$f', round(d['ms_per_step'],4), round(d['value'],1), (d.get('roofline') or {}).get('frac'), (d.get('unpipelined') or {}).get('ms_per_step'))"; done
head -8 $O/encoder_kernel_stats.txt | cut -c1-150
