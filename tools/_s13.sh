mkdir -p gpurun_out/s13
for rep in 1 2; do
for v in "8:128" "8:96" "8:160" "8:192" "12:512" "12:384" "12:768" "5:512" "5:384" "5:768" "7:384" "7:256" "7:512" "14:1024" "14:768" "14:1536" "16:1024" "16:512" "16:1536"; do
  ISTNET_PW_TUNE=$v python bench.py --no-roofline --no-cpu-baseline --no-eager-leg --no-unpipelined --steps 50 --warmup 10 --windows 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],4), d.get('windows_ms_per_step'))"
done; done > gpurun_out/s13/sweep.txt 2>&1
cat gpurun_out/s13/sweep.txt
