#!/bin/bash
# round 5 (second half): the measurements that go to profiles/ (run from the repo root on the GPU box)
O=gpurun_out/r05f; mkdir -p $O
bash tools/pmc_traffic.sh r05f > $O/pmc_traffic.txt 2>&1
cp gpurun_out/pmc_r05f_traffic.json $O/pmc_traffic.json; cp $O/pmc_traffic.json profiles/r05_pmc_traffic.json    # bench.py reads it (source hash checked)
rm -rf gpurun_out/pmc_r05f_FETCH_SIZE gpurun_out/pmc_r05f_WRITE_SIZE
for mode in "" "--no-prefetch"; do
  tag=pipelined; [ -n "$mode" ] && tag=unpipelined
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_r05f_$tag -o enc -- python /root/repo/bench.py --no-roofline --no-cpu-baseline --no-unpipelined --no-eager-leg --steps 20 --warmup 5 --windows 1 $mode > /root/repo/$O/prof_$tag.log 2>&1)
  python tools/rocprof_summary.py gpurun_out/prof_r05f_$tag/enc_results.db > $O/encoder_kernel_stats_$tag.txt
  python tools/step_kernel_list.py gpurun_out/prof_r05f_$tag/enc_results.db 0 > $O/step_kernel_timeline_$tag.txt
  rm -rf gpurun_out/prof_r05f_$tag
done
python bench.py 2>$O/bench_final.err | tail -1 > $O/bench_final.json
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_form.json
python bench.py --no-prefetch --no-roofline --no-cpu-baseline --no-eager-leg --steps 50 --warmup 10 2>/dev/null | tail -1 > $O/bench_noprefetch.json
python bench.py --workload sa_layer --steps 200 --warmup 20 2>/dev/null | tail -1 > $O/bench_sa_layer.json
python bench.py --workload istnet --no-roofline --steps 30 --warmup 5 2>/dev/null | tail -1 > $O/bench_istnet_full_model.json
python bench.py --workload infer --no-roofline --steps 30 --warmup 5 2>/dev/null | tail -1 > $O/bench_infer_full_model.json
python bench.py --workload pipeline --no-roofline --steps 30 --warmup 5 2>/dev/null | tail -1 > $O/bench_pipeline.json
python tools/step_timeline.py 2>&1 | grep -v amdgpu.ids > $O/step_timeline.txt
python tools/gemm_launch_table.py 2>&1 | grep -v amdgpu.ids > $O/encoder_gemm_launch_table.txt
for f in bench_final bench_driver_form bench_noprefetch bench_sa_layer bench_istnet_full_model bench_infer_full_model bench_pipeline; do python -c "
import json; d=json.load(open('$O/$f.json')); print('$f', round(d['ms_per_step'],4), round(d['value'],1), (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('traffic'), (d.get('unpipelined') or {}).get('ms_per_step'))"; done
head -6 $O/encoder_kernel_stats_pipelined.txt | cut -c1-160
