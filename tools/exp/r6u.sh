#!/bin/bash
# slim evidence refresh after a kernel edit: PMC summaries (hash-tied to the sources), kernel stats, the committed bench lines
O=gpurun_out/r6u; mkdir -p $O
tools/pmc_sq.sh r6u > $O/pmc_sq.out 2>&1
tools/pmc_traffic.sh r6u > $O/pmc_traffic.out 2>&1
cp gpurun_out/pmc_r6u_sq.json profiles/r06_pmc_sq_counters.json; cp gpurun_out/pmc_r6u_traffic.json profiles/r06_pmc_traffic.json
tools/gpu_session.sh r6u prof > $O/session.txt 2>&1
tools/gpu_session.sh r6u_np prof:--no-prefetch >> $O/session.txt 2>&1
for i in 1 2 3; do python bench.py > $O/bench_$i.json 2>/dev/null; done
python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2>/dev/null
python -m pytest tests -m gpu -q -W ignore 2>&1 | tail -2
python - <<PY
import json
for f in ("bench_1", "bench_2", "bench_3", "bench_driver_form"):
    d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f, round(d["ms_per_step"], 4), "unpipelined", round(d["unpipelined"]["ms_per_step"], 4), "frac", round(r["frac"], 3), "traffic", r["traffic"], "mfma", r.get("mfma_busy_frac"), "eager", round(d["eager"]["graph_segments/torch.optim.Adam"]["ms_per_step"], 3), {k: round(v["ms_per_step"], 3) for k, v in d["other_distributions"].items() if isinstance(v, dict)})
PY
head -3 gpurun_out/r6u/kernel_stats_1.txt; head -3 gpurun_out/r6u_np/kernel_stats_1.txt
