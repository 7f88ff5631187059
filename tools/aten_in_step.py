"""Which framework (ATen / runtime) kernels are still inside an encoder training step, and which Python line issues
them: one eager step under torch.profiler with stacks; kernels of libistnet_pn2.so are summarised by count only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from istnet_amd.optim import FlatAdam
from istnet_amd.modules import GeometrySlot

dev = torch.device("cuda:0")
workload = sys.argv[1] if len(sys.argv) > 1 else "encoder"
if workload == "encoder":
    model = bench.make_model(dev)
    batches = [bench.shell_cloud(32, 1024, s, dev) for s in (0, 1000)]
    slots = [model.prefetch_geometry(bt, GeometrySlot()) for bt in batches]
    fwd = [bench.make_pipelined_fwd_bwd(model, batches, slots, i) for i in (0, 1)]
else:
    model = bench.make_istnet(dev)
    fwd = [bench.make_istnet_fwd_bwd(model, bench.istnet_batch(32, 1024, 0, dev))]
opt = FlatAdam(model.parameters(), lr=1e-4)
step = bench.make_eager_step(fwd, opt, 1)
for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ours, other = 0, {}
for ev in prof.events():
    if ev.device_type.name != "CUDA":
        continue
    name = ev.name
    if "anonymous namespace" in name and "at::native" not in name:
        ours += 1
        continue
    other.setdefault(name[:110], []).append(ev)
print(f"{workload}: {ours} launches of the library's kernels, {sum(len(v) for v in other.values())} other device activities")
# attribute by the CPU op that launched: walk CPU events with stacks
cpu = [e for e in prof.events() if e.device_type.name == "CPU" and e.stack]
for name, evs in sorted(other.items(), key=lambda kv: -len(kv[1])):
    print(f"{len(evs):4d} x {name}")
print("---- CPU ops that launch framework kernels (aten::*), by source line ----")
seen = {}
for e in cpu:
    if not e.name.startswith("aten::") or e.name in ("aten::empty", "aten::view", "aten::as_strided", "aten::empty_strided",
                                                        "aten::reshape", "aten::slice", "aten::select", "aten::transpose",
                                                        "aten::t", "aten::expand", "aten::unsqueeze", "aten::squeeze",
                                                        "aten::_unsafe_view", "aten::detach", "aten::alias", "aten::permute",
                                                        "aten::empty_like", "aten::contiguous", "aten::result_type", "aten::to",
                                                        "aten::view_as", "aten::narrow", "aten::resize_", "aten::lift_fresh",
                                                        "aten::unbind", "aten::item", "aten::_local_scalar_dense", "aten::is_nonzero"):
        continue
    frames = [f for f in e.stack if "/root/repo" in f or "bench.py" in f or "istnet" in f]
    key = (e.name, frames[0] if frames else (e.stack[0] if e.stack else "?"))
    seen[key] = seen.get(key, 0) + 1
for (name, where), n in sorted(seen.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{n:4d} x {name:28s} {where}")
