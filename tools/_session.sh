run() { echo -n "$1: "; env $1 python bench.py --no-roofline --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['unpipelined']['ms_per_step'],4))"; }
run "A=1"
run "ISTNET_EXP_SKIP_DEFERRED_WGRAD=1"
run "ISTNET_DEFERRED_WGRAD=0"
run "A=1"
run "ISTNET_EXP_SKIP_DEFERRED_WGRAD=1"
