mkdir -p gpurun_out/r4c
python -m pytest tests -m gpu -x -q 2>&1 | grep -v "Warn\|warn" | tail -30 > gpurun_out/r4c/test.txt; grep -n "Error\|^E \|passed\|failed" gpurun_out/r4c/test.txt | head -20
python bench.py --no-roofline --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], (d.get('unpipelined') or {}).get('ms_per_step'))"
