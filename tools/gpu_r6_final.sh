#!/bin/bash
# Round-6 evidence session (GPU box, repo root): everything profiles/r06_* is made from, on ONE box at HEAD.
#   tools/gpu_r6_final.sh [tag]   -> gpurun_out/<tag>/...
TAG=${1:-r6final}; O=gpurun_out/$TAG; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/test.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
python bench.py > $O/bench_final.json 2> $O/bench_final.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2>/dev/null
python bench.py --no-prefetch --no-cpu-baseline --no-eager-leg --no-roofline > $O/bench_noprefetch.json 2>/dev/null
for c in cube dense; do
  python bench.py --cloud $c --no-cpu-baseline --no-eager-leg > $O/bench_$c.json 2>/dev/null
done
tools/gpu_session.sh $TAG prof "py:gemm_launch_table.py" "py:step_timeline.py" > $O/session.txt 2>&1
PROF_STEPS=29 tools/gpu_session.sh ${TAG}_np prof:--no-prefetch >> $O/session.txt 2>&1
tools/pmc_sq.sh $TAG > $O/pmc_sq.out 2>&1
tools/pmc_traffic.sh $TAG > $O/pmc_traffic.out 2>&1
for w in sa_layer istnet infer; do
  python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2>/dev/null
done
tail -3 $O/test.txt; tail -2 $O/smoke.txt
python - <<PY
import json
for f in ("bench_final", "bench_driver_form", "bench_noprefetch", "bench_cube", "bench_dense", "bench_sa_layer", "bench_istnet", "bench_infer"):
    try:
        d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print(f, round(d["ms_per_step"], 4), "ms", (d.get("unpipelined") or {}).get("ms_per_step"), "frac", r.get("frac"), "all_gemm", r.get("all_gemm_kernels_frac"),
              "fam", (r.get("dominant_family") or {}).get("frac"), "mfma_busy", r.get("mfma_busy_frac"))
    except Exception as e:
        print(f, "failed", e)
PY
