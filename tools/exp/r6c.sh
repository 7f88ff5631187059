#!/bin/bash
# round-6 session c: forward 64 x 128 split-K tiles (dgrad back at 32 x 128), split prefetch, 4-wave FPS in the un-pipelined step
O=gpurun_out/r6c; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/test.txt
python tools/bench_sk.py > $O/bench_sk.txt 2>&1
ab() {  # ab <label> <env assignments...>
  label=$1; shift
  for extra in "" "--no-prefetch"; do
    env "$@" python bench.py --no-roofline --no-cpu-baseline --no-eager-leg --no-other-clouds --no-unpipelined --steps 50 --warmup 10 --windows 3 $extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', '$extra', round(d['ms_per_step'],4), d.get('windows_ms_per_step'))"
  done
}
for rep in 1 2; do
  ab "tm1          " ISTNET_PW_TUNE=24:0
  ab "tm2(default) " ISTNET_PW_TUNE=24:256
  ab "split=after_sa" ISTNET_PREFETCH_SPLIT=after_sa
  ab "split=after_fwd" ISTNET_PREFETCH_SPLIT=after_fwd
  ab "fps 4 waves  " ISTNET_PN2_TUNE=0:1024
done > $O/ab.txt 2>&1
tail -3 $O/test.txt; cat $O/ab.txt; tail -14 $O/bench_sk.txt
