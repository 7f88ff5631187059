"""Every GEMM launch of an encoder training step by (kernel, algorithmic flop, bytes): count per step, average time from
HIP events around the launch in instrumented eager steps, rate against the fp32 MFMA roof and the HBM roof."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from istnet_amd import _native
from istnet_amd.optim import FlatAdam, layout_hints
from istnet_amd.modules import GeometrySlot

dev = torch.device("cuda:0")
model = bench.make_model(dev)
batches = [bench.shell_cloud(32, 1024, s, dev) for s in (0, 1000)]
slots = [model.prefetch_geometry(bt, GeometrySlot()) for bt in batches]
fwd = [bench.make_pipelined_fwd_bwd(model, batches, slots, i) for i in (0, 1)]
opt = FlatAdam(model.parameters(), lr=1e-4, adjacent=layout_hints(model))
step = bench.make_eager_step(fwd, opt, 1)
for _ in range(4):
    step()
torch.cuda.synchronize()
STEPS = 6
_native.TIMING = []
for _ in range(STEPS):
    step()
torch.cuda.synchronize()
rec, _native.TIMING = _native.TIMING, None
agg = {}
for name, flops, nbytes, s, e in rec:
    k = (name.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", ""), flops, nbytes)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += s.elapsed_time(e) * 1e3
tot = sum(a[1] for a in agg.values()) / STEPS
print(f"# {len(rec) // STEPS} timed GEMM launches per step, {tot:.0f} us of launch time per step (eager, HIP events)")
print("  us/step  n/step   avg_us   GFLOP     MB    TF/s  frac_mfma   GB/s  kernel")
for (name, flops, nbytes), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    avg = us / n
    print(f"{us / STEPS:9.1f} {n / STEPS:7.1f} {avg:8.1f} {flops / 1e9:7.3f} {nbytes / 1e6:6.1f} {flops / avg / 1e6:7.1f} {flops / avg / 1e6 / 157.3:10.3f} "
          f"{nbytes / avg / 1e3:6.0f}  {name[:60]}")
