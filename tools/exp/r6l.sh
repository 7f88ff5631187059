#!/bin/bash
# round-6 session l: the committed bench lines again at HEAD (adaptive compact columns in)
O=gpurun_out/r6l; mkdir -p $O
python bench.py > $O/bench_final.json 2> $O/bench_final.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2>/dev/null
for c in cube dense; do
  python bench.py --cloud $c --no-cpu-baseline --no-eager-leg > $O/bench_$c.json 2>/dev/null
done
python -m pytest tests -m gpu -q -W ignore 2>&1 | tail -3 > $O/test.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
tail -2 $O/test.txt; tail -1 $O/smoke.txt
python - <<PY
import json
for f in ("bench_final", "bench_driver_form", "bench_cube", "bench_dense"):
    d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
    print(f, round(d["ms_per_step"], 4), d["windows_ms_per_step"], "unpipelined", (d.get("unpipelined") or {}).get("ms_per_step"), "frac", r.get("frac"), "traffic", r.get("traffic"), "mfma_busy", r.get("mfma_busy_frac"),
          {k: (v.get("ms_per_step"), v.get("compact_columns_level1")) for k, v in (d.get("other_distributions") or {}).items() if isinstance(v, dict)})
PY
