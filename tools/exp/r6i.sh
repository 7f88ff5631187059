#!/bin/bash
# round-6 session i: kernel-trace stats with the extra legs off, final bench line with the PMC summaries in place, GPU tests in reverse file order
O=gpurun_out/r6i; mkdir -p $O
tools/gpu_session.sh r6i prof > $O/session.txt 2>&1
tools/gpu_session.sh r6i_np prof:--no-prefetch >> $O/session.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d /root/repo/$O/tl -o enc -- python /root/repo/bench.py --no-roofline --no-cpu-baseline --no-unpipelined --no-other-clouds --no-eager-leg --steps 20 --warmup 5 > /root/repo/$O/tl.log 2>&1)
python tools/step_kernel_list.py $O/tl/enc_results.db 0 > $O/step_kernel_timeline.txt 2>&1; rm -rf $O/tl
python bench.py > $O/bench_final.json 2> $O/bench_final.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2>/dev/null
python -m pytest $(ls tests/test_*.py | tac) -m gpu -q -p no:cacheprovider 2>&1 | tail -6 > $O/test_reverse.txt
tail -3 $O/test_reverse.txt
python - <<PY
import json
for f in ("bench_final", "bench_driver_form"):
    d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f, round(d["ms_per_step"], 4), d["windows_ms_per_step"], "unpipelined", d["unpipelined"]["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"], "mfma_busy", r.get("mfma_busy_frac"))
PY
head -3 gpurun_out/r6i/kernel_stats_1.txt; head -3 gpurun_out/r6i_np/kernel_stats_1.txt
