#!/bin/bash
# round-6 session h: which fork / join sites of the step pay for themselves (ISTNET_NO_FORK: the named sites stay on one stream)
O=gpurun_out/r6h; mkdir -p $O
run() {
  label=$1; shift
  for extra in "" "--no-prefetch"; do
  env "$@" python bench.py --no-roofline --no-cpu-baseline --no-eager-leg --no-other-clouds --no-unpipelined --steps 50 --warmup 10 --windows 3 $extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label $extra', round(d['ms_per_step'],4), d.get('windows_ms_per_step'))"
  done
}
{
run "base" X=1
run "no fpskip" ISTNET_NO_FORK=fpskip
run "no fwd512" ISTNET_NO_FORK=fwd512
run "no bwd512" ISTNET_NO_FORK=bwd512
run "no fwd512,bwd512" ISTNET_NO_FORK=fwd512,bwd512
run "base" X=1
run "no fwd256" ISTNET_NO_FORK=fwd256
run "no bwd256" ISTNET_NO_FORK=bwd256
run "no fwd128" ISTNET_NO_FORK=fwd128
run "no bwd128" ISTNET_NO_FORK=bwd128
run "no fwd64" ISTNET_NO_FORK=fwd64
run "no bwd64" ISTNET_NO_FORK=bwd64
run "base" X=1
run "fp bwd_mid wgs 128" ISTNET_FP_BWD_MID_WGS=128
run "fp bwd_mid wgs 192" ISTNET_FP_BWD_MID_WGS=192
run "compact levels 0,1" ISTNET_COMPACT_LEVELS=0,1
run "base" X=1
} > $O/sites.txt 2>&1
cat $O/sites.txt
