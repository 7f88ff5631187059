#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_pw_last_gpu.py -q 2>&1 | tail -5
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_r3g -o last -- python /root/repo/tools/bench_last.py > /dev/null 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_r3g/last_results.db | grep -E "pw_bwd_last|pw_bwd_mid|pw_dw_last|pw_last_prep|pw_fwd2|pool|wgrad2_kernel<128, 128, true" | cut -c1-150
rm -rf gpurun_out/prof_r3g
