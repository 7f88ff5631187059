#!/bin/bash
# usage (GPU box, repo root): tools/pmc_few.sh <tag>   -- SQ counters of the GEMM kernels on a few layer shapes
TAG=${1:-x}
mkdir -p gpurun_out
i=0
for C in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc $C -d /root/repo/gpurun_out/pmcf_${TAG}_$i -o p -- python /root/repo/tools/exp/bench_pw_few.py > /root/repo/gpurun_out/pmcf_${TAG}_$i.log 2>&1)
  echo "== pass $i: $C"
  python tools/pmc_summary.py gpurun_out/pmcf_${TAG}_$i/p_results.db pw_
  rm -rf gpurun_out/pmcf_${TAG}_$i
done
