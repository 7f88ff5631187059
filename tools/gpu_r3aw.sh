#!/bin/bash
run() { python bench.py --workload istnet --no-roofline --steps 20 --warmup 5 $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), d['config']['launch'])"; }
run "new default" ""
run "new default" ""
ISTNET_EARLY_WORLD=0 run "new default, no early world" ""
ISTNET_FP_SKIP_STREAM=0 run "new default, no fp skip" ""
ISTNET_GEOMETRY_STREAM=0 run "new default, no geometry stream" ""
run "new default eager" "--eager"
python bench.py --workload istnet --force-dist --no-roofline --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('force-dist', round(d['ms_per_step'],3), d['config']['launch'])"
timeout 600 python -m pytest tests/test_golden_gpu.py -q -x -k "istnet or supervised or config_3 or full" 2>&1 | tail -2
