#!/bin/bash
# round-6 session d: re-sweep of the launch-size knobs on HEAD's kernels (pipelined step, 3 windows of 50 steps each)
O=gpurun_out/r6d; mkdir -p $O
run() {
  label=$1; shift
  env "$@" python bench.py --no-roofline --no-cpu-baseline --no-eager-leg --no-other-clouds --no-unpipelined --steps 50 --warmup 10 --windows 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['ms_per_step'],4), d.get('windows_ms_per_step'))"
}
{
run "base" X=1
for v in 96 160 192 256; do run "bwd_mid_target(8)=$v" ISTNET_PW_TUNE=8:$v; done
for v in 256 384 768; do run "wgrad2_target(12)=$v" ISTNET_PW_TUNE=12:$v; done
for v in 256 768; do run "bwd_small_target(5)=$v" ISTNET_PW_TUNE=5:$v; done
run "base" X=1
for v in 512 2048 4096; do run "fwd_sk_max_tiles(16)=$v" ISTNET_PW_TUNE=16:$v; done
for v in 512 2048; do run "fwd2_min_waves(14)=$v" ISTNET_PW_TUNE=14:$v; done
for v in 128 512; do run "dgrad_sk_min_k(18)=$v" ISTNET_PW_TUNE=18:$v; done
for v in 256 512; do run "dgrad_min_wgs(7)=$v" ISTNET_PW_TUNE=7:$v; done
for v in 384 768; do run "wg_target_big(1)=$v" ISTNET_PW_TUNE=1:$v; done
for v in 384 768; do run "wg_target_small(2)=$v" ISTNET_PW_TUNE=2:$v; done
run "base" X=1
run "wgrad2 tiles 128 (20)" ISTNET_PW_TUNE=20:128
run "wgrad2 nt 128 (23)" ISTNET_PW_TUNE=23:128
run "scatter threads 512 (21)" ISTNET_PW_TUNE=21:512
run "sk_tm2 min 512 (24)" ISTNET_PW_TUNE=24:512
run "sk_tm2 min 128 (24)" ISTNET_PW_TUNE=24:128
run "base" X=1
} > $O/sweep.txt 2>&1
cat $O/sweep.txt
