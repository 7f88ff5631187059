for rep in 1 2; do
for cfg in "" "ISTNET_EXP_SPLIT_GEOM=1" "ISTNET_EXP_SPLIT_GEOM=1 ISTNET_EXP_PRIO=1" "ISTNET_EXP_PRIO=1"; do
  env $cfg python bench.py --no-roofline --no-cpu-baseline --no-eager-leg --no-unpipelined --steps 50 --warmup 10 --windows 3 2>&1 | tail -1 | python -c "import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$cfg]', round(d['ms_per_step'],4), d.get('windows_ms_per_step'))
except Exception as e: print('[$cfg] failed', e)"
done; done
