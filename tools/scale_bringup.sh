#!/bin/bash
# Bring-up order for the FIRST run of the RCCL path on more than one device (no multi-GPU node has been available to this
# repo in six rounds: RCCL has never seen two ranks of this code, a graph capture has never run under the NCCL watchdog with
# peers).  Run on an N-GPU MI355X node from the repo root:   tools/scale_bringup.sh [NMAX=8]
# Every stage has its own timeout and log under gpurun_out/scale/; a failing stage names the fallback the next stages (and a
# user) should take.  DESIGN.md section 8 holds the prediction these numbers are to be held against.  No number is claimed here.
NMAX=${1:-8}
O=gpurun_out/scale; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0          # the host driver supports dmabuf IPC only (hipIpcGetMemHandle fails otherwise)
COMMON="--no-cpu-baseline --no-roofline --no-eager-leg --no-other-clouds --steps 20 --warmup 5"
run() {  # run <name> <n> <extra bench args...>
  name=$1; n=$2; shift 2
  port=$((29500 + RANDOM % 2000))
  echo "== $name (N=$n): bench.py --gpus $n $*"
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
      bench.py --gpus $n $COMMON "$@" > $O/$name.json 2> $O/$name.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "   FAILED rc=$rc (tail of $O/$name.err):"; tail -5 $O/$name.err; return 1; fi
  python - <<PY
import json
d = json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("   %.3f ms/step  %.0f clouds/s  launch=%s  exchange=%s" % (d["ms_per_step"], d["value"], d["config"]["launch"],
      (d["config"].get("gradient_exchange") or {}).get("issued")))
PY
}
# 0. one rank through the N>1 code path (what the 1-GPU boxes have run since round 3): separates RCCL-with-peers problems
#    from problems of the path itself
run n1_force_dist 1 --force-dist || echo "   -> the path itself is broken on this box: stop here"
# 1. two ranks, EAGER step, buckets from the autograd hooks; NCCL_DEBUG shows the ranks and the transport RCCL picked
NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,GRAPH run n2_eager 2 --eager --overlap-allreduce \
  || echo "   -> RCCL cannot form a 2-rank communicator: check HSA_ENABLE_IPC_MODE_LEGACY=0, ulimit -l, /dev/kfd permissions; nothing below can work"
grep -h "NCCL INFO.*\(Channel\|via\|Connected\|nranks\|comm \)" $O/n2_eager.err | head -20
# 2. two ranks, the DEFAULT form: captured step + the buckets back to back after the replay (no collective inside a capture)
run n2_graph 2 || echo "   -> capture next to a live communicator fails: bench.py falls back to the eager step by itself; use --eager"
# 3. two ranks, collectives INSIDE the captured step (opt-in until this stage has passed once)
run n2_captured 2 --capture-allreduce \
  || echo "   -> RCCL capture with peers fails (watchdog / stream capture mode): keep the default form (stage 2); bench.py already falls back"
# 4. the node
for n in 4 $NMAX; do
  [ $n -le $NMAX ] || continue
  run n${n}_graph $n || echo "   -> N=$n default form failed although N=2 passed: topology / link problem, see NCCL_DEBUG=INFO"
  run n${n}_captured $n --capture-allreduce || echo "   -> keep the default form at N=$n"
done
# 5. the full model (107 MB exchange in ~4 buckets) in the default form
run istnet_n${NMAX}_graph $NMAX --workload istnet || echo "   -> full-model default form failed at N=$NMAX"
run istnet_n${NMAX}_overlap $NMAX --workload istnet --overlap-allreduce || true
echo "logs: $O/"
