#!/bin/bash
# usage (GPU box, repo root): tools/pmc_sq.sh <tag>
# SQ / GRBM counters of every kernel of the encoder step (bench.py --eager: the real step, launch by launch), each counter
# group in its OWN rocprofv3 --pmc pass with --kernel-trace only (no other trace domain).  Writes
#   gpurun_out/pmc_<tag>_sq.txt   per-kernel table
#   gpurun_out/pmc_<tag>_sq.json  what roofline.pmc_sq reads (copy to profiles/r06_pmc_sq_counters.json)
TAG=${1:-x}
mkdir -p gpurun_out
PASS1="SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
PASS2="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES SQ_BUSY_CYCLES"
i=0
for C in "$PASS1" "$PASS2"; do
  i=$((i+1))
  (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --pmc $C -d /root/repo/gpurun_out/pmcsq_${TAG}_$i -o p -- python /root/repo/bench.py --eager --no-roofline --no-cpu-baseline --no-eager-leg --no-other-clouds --no-unpipelined --steps 3 --warmup 2 > /root/repo/gpurun_out/pmcsq_${TAG}_$i.log 2>&1)
  echo "pass $i rc=$?"
done
python tools/pmc_sq_summary.py gpurun_out/pmcsq_${TAG}_1/p_results.db gpurun_out/pmcsq_${TAG}_2/p_results.db gpurun_out/pmc_${TAG}_sq
rm -rf gpurun_out/pmcsq_${TAG}_1 gpurun_out/pmcsq_${TAG}_2      # the databases are tens of MB: gpurun copies back <= 64 MiB
