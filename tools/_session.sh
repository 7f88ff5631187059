run() { echo -n "$1: "; env $1 python bench.py --workload istnet --no-roofline --steps 20 --warmup 5 2>/tmp/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))" || tail -5 /tmp/err.txt; }
run "ISTNET_RGB_LAST=1"; run "ISTNET_RGB_LAST=0"; run "ISTNET_RGB_LAST=1"
ISTNET_SCALE_STREAMS=0 ISTNET_DEFERRED_WGRAD=0 python tools/istnet_step_timeline.py 2>/dev/null | head -60
