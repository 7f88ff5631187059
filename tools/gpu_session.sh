#!/bin/bash
# One parameterised GPU-box session script (replaces the per-experiment one-offs of earlier rounds).
# usage (GPU box, repo root):  tools/gpu_session.sh <tag> <step> [<step> ...]
# steps:
#   test[:expr]      pytest -m gpu (optionally -k expr)
#   bench[:args]     bench.py line -> gpurun_out/<tag>/bench_<n>.json      (args: comma separated bench flags)
#   prof[:args]      rocprofv3 --kernel-trace --stats of bench.py -> <tag>/kernel_stats_<n>.txt
#   ab[:args]        A/B of the working-tree library against ab_base/base.so (built beforehand in the container
#                    with tools/build_base.sh <rev>; ab_base/ is git-ignored and travels with the snapshot)
#   py:<script>[:args]  python tools/<script> -> <tag>/<script>_<n>.txt
TAG=${1:-x}; shift
O=gpurun_out/$TAG; mkdir -p $O
n=0
for step in "$@"; do
  n=$((n+1))
  kind=${step%%:*}; rest=${step#*:}; [ "$rest" = "$step" ] && rest=""
  case $kind in
    test)
      if [ -n "$rest" ]; then python -m pytest tests -m gpu -x -q -k "$rest" 2>&1 | tail -8
      else python -m pytest tests -m gpu -x -q 2>&1 | tail -8; fi | tee $O/test_$n.txt ;;
    bench)
      args=${rest//,/ }
      python bench.py $args 2>$O/bench_$n.err | tail -1 > $O/bench_$n.json
      python - <<PY
import json
try:
    d = json.load(open('$O/bench_$n.json'))
    r = d.get('roofline') or {}
    print('BENCH[$n] $rest', round(d['ms_per_step'], 4), 'ms', round(d['value'], 1), d['unit'], 'frac', r.get('frac'),
          'unpipelined', (d.get('unpipelined') or {}).get('ms_per_step'))
except Exception as e:
    print('BENCH[$n] failed', e); print(open('$O/bench_$n.err').read()[-2000:])
PY
      ;;
    prof)
      args=${rest//,/ }
      (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof_$n -o enc -- \
         python /root/repo/bench.py --no-roofline --no-cpu-baseline --no-unpipelined --no-other-clouds --no-eager-leg --steps 20 --warmup 5 $args \
         > /root/repo/$O/prof_$n.log 2>&1)
      python tools/rocprof_summary.py $O/prof_$n/enc_results.db ${PROF_STEPS:-29} > $O/kernel_stats_$n.txt
      rm -rf $O/prof_$n
      head -${LINES_SHOWN:-30} $O/kernel_stats_$n.txt | cut -c1-150 ;;
    ab)
      args=${rest//,/ }
      bash tools/ab.sh 2 $args 2>&1 | tee $O/ab_$n.txt ;;
    py)
      script=${rest%%:*}; pargs=${rest#*:}; [ "$pargs" = "$rest" ] && pargs=""
      python tools/$script ${pargs//,/ } 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tee $O/${script%.py}_$n.txt | tail -${LINES_SHOWN:-40} ;;
    *) echo "unknown step $step" ;;
  esac
done
