#!/bin/bash
# round-6 session e: finer sweep around the two knobs the re-sweep found (key 1: split-K workgroups of pw_wgrad_kernel, big outputs;
# key 14: wave-tile count from which a forward launch takes pw_fwd2_kernel)
O=gpurun_out/r6e; mkdir -p $O
run() {
  label=$1; shift
  env "$@" python bench.py --no-roofline --no-cpu-baseline --no-eager-leg --no-other-clouds --no-unpipelined --steps 50 --warmup 10 --windows 3 $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label $EXTRA', round(d['ms_per_step'],4), d.get('windows_ms_per_step'))"
}
{
run "base" X=1
for v in 640 768 896 1024 1280 1536 2048; do run "k1=$v" ISTNET_PW_TUNE=1:$v; done
run "base" X=1
for v in 1536 2048 3072 4096 8192 1000000; do run "k14=$v" ISTNET_PW_TUNE=14:$v; done
run "base" X=1
for a in 768 1024; do for b in 2048 4096; do run "k1=$a,k14=$b" ISTNET_PW_TUNE=1:$a,14:$b; done; done
for a in 768 1024; do for c in 640 768 1024; do run "k1=$a,k2=$c" ISTNET_PW_TUNE=1:$a,2:$c; done; done
run "base" X=1
EXTRA=--no-prefetch
run "base" X=1
run "k1=768" ISTNET_PW_TUNE=1:768
run "k14=2048" ISTNET_PW_TUNE=14:2048
run "k1=768,k14=2048" ISTNET_PW_TUNE=1:768,14:2048
run "k1=1024,k14=4096" ISTNET_PW_TUNE=1:1024,14:4096
run "base" X=1
} > $O/sweep.txt 2>&1
cat $O/sweep.txt
