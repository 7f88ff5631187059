#!/bin/bash
O=gpurun_out/r6p; mkdir -p $O
for i in 1 2 3; do python bench.py > $O/bench_$i.json 2>/dev/null; done
python - <<PY
import json
for i in (1,2,3):
    d = json.loads(open("$O/bench_%d.json" % i).read().strip().splitlines()[-1]); r = d["roofline"]
    print(i, round(d["ms_per_step"], 4), "unpipelined", round(d["unpipelined"]["ms_per_step"],4), "frac", round(r["frac"],3), "eager", round(d["eager"]["graph_segments/torch.optim.Adam"]["ms_per_step"],3), {k: round(v["ms_per_step"],3) for k,v in d["other_distributions"].items() if isinstance(v, dict)})
PY
