#!/bin/bash
# round-6 session o: everything that is tied to the kernel sources, again at HEAD (4-wave FPS default)
O=gpurun_out/${TAG:-r6o}; mkdir -p $O
python -m pytest tests -m gpu -q -W ignore 2>&1 | tail -3 > $O/test.txt
tools/pmc_sq.sh ${TAG:-r6o} > $O/pmc_sq.out 2>&1
tools/pmc_traffic.sh ${TAG:-r6o} > $O/pmc_traffic.out 2>&1
cp gpurun_out/pmc_${TAG:-r6o}_sq.json profiles/r06_pmc_sq_counters.json; cp gpurun_out/pmc_${TAG:-r6o}_traffic.json profiles/r06_pmc_traffic.json
python bench.py > $O/bench_final.json 2> $O/bench_final.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2>/dev/null
python bench.py --no-prefetch --no-cpu-baseline --no-eager-leg --no-roofline > $O/bench_noprefetch.json 2>/dev/null
for c in cube dense; do python bench.py --cloud $c --no-cpu-baseline --no-eager-leg > $O/bench_$c.json 2>/dev/null; done
tools/gpu_session.sh ${TAG:-r6o} prof "py:step_timeline.py" "py:gemm_launch_table.py" > $O/session.txt 2>&1
tools/gpu_session.sh ${TAG:-r6o}_np prof:--no-prefetch >> $O/session.txt 2>&1
python tools/bench_infer_small.py > $O/infer_small.txt 2>&1
python tools/bench_fps_chain.py > $O/fps_chain.txt 2>&1
tail -2 $O/test.txt
python - <<PY
import json
for f in ("bench_final", "bench_driver_form", "bench_noprefetch", "bench_cube", "bench_dense"):
    d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
    print(f, round(d["ms_per_step"], 4), d["windows_ms_per_step"], "unpipelined", (d.get("unpipelined") or {}).get("ms_per_step"), "frac", r.get("frac"), "traffic", r.get("traffic"), "mfma_busy", r.get("mfma_busy_frac"),
          "eager", ((d.get("eager") or {}).get("graph_segments/torch.optim.Adam") or {}).get("ms_per_step"))
PY
head -3 gpurun_out/${TAG:-r6o}/kernel_stats_1.txt; head -3 gpurun_out/${TAG:-r6o}_np/kernel_stats_1.txt; grep -v amdgpu $O/infer_small.txt | tail -8; grep -v amdgpu $O/fps_chain.txt | tail -3
