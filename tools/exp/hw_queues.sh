for q in ${QS:-1 2 3 4 5}; do
  for extra in "" "--no-prefetch"; do
    GPU_MAX_HW_QUEUES=$q python bench.py --no-roofline --no-cpu-baseline --no-eager-leg --steps 50 --warmup 10 --windows 3 $extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('q=$q', '$extra', round(d['ms_per_step'],4), d.get('windows_ms_per_step'))"
  done
done
