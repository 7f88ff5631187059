#!/bin/bash
# round-6 session b: split-K 64 x 128 tiles -- microbench, parity tests, A/B of the step, full line, counters
O=gpurun_out/r6b; mkdir -p $O
python tools/bench_sk.py > $O/bench_sk.txt 2>&1
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/test.txt
for rep in 1 2; do
  for v in "24:0" "24:256"; do
    for extra in "" "--no-prefetch"; do
      ISTNET_PW_TUNE=$v python bench.py --no-roofline --no-cpu-baseline --no-eager-leg --no-other-clouds --no-unpipelined --steps 50 --warmup 10 --windows 3 $extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TUNE=$v', '$extra', round(d['ms_per_step'],4), d.get('windows_ms_per_step'))"
    done
  done
done > $O/ab.txt 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
tools/pmc_sq.sh r6b > $O/pmc_sq.out 2>&1
tools/pmc_traffic.sh r6b > $O/pmc_traffic.out 2>&1
tail -3 $O/test.txt; cat $O/ab.txt; tail -3 $O/bench_sk.txt
