"""Which gradients of the pipelined encoder step are not written in place into FlatAdam's buffer, and which device-to-device copies /
framework kernels a step issues (torch profiler, every event with its Python stack)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from istnet_amd.optim import FlatAdam, layout_hints
from istnet_amd.modules import GeometrySlot
dev = torch.device("cuda:0")
model = bench.make_model(dev)
batches = [bench.shell_cloud(32, 1024, s, dev) for s in (0, 1000)]
slots = [model.prefetch_geometry(bt, GeometrySlot()) for bt in batches]
fwd = [bench.make_pipelined_fwd_bwd(model, batches, slots, i) for i in (0, 1)]
opt = FlatAdam(model.parameters(), lr=1e-4, adjacent=layout_hints(model))
names = {id(p): n for n, p in model.named_parameters()}
opt.zero_grad(set_to_none=True)
fwd[0]()
torch.cuda.synchronize()
for p in opt.params:
    slot = p._istnet_grad_slot
    if p.grad is None:
        print("NO GRAD     ", names[id(p)], tuple(p.shape))
    elif p.grad.data_ptr() != slot.data_ptr() or not p.grad.is_contiguous():
        print("NOT IN PLACE", names[id(p)], tuple(p.shape), p.grad.is_contiguous())
opt.step()
step = bench.make_eager_step(fwd, opt, 1)
for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
seen = {}
for ev in prof.events():
    if ev.device_type.name == "CPU" and (ev.name.startswith("aten::") or "emcpy" in ev.name or "emset" in ev.name):
        if ev.name in ("aten::empty", "aten::empty_strided", "aten::view", "aten::as_strided", "aten::slice", "aten::select", "aten::empty_like",
                       "aten::reshape", "aten::transpose", "aten::unsqueeze", "aten::squeeze", "aten::detach", "aten::alias", "aten::narrow",
                       "aten::_unsafe_view", "aten::expand", "aten::permute", "aten::t", "aten::result_type", "aten::to", "aten::lift_fresh",
                       "aten::contiguous", "aten::is_nonzero", "aten::item", "aten::_local_scalar_dense", "aten::unbind", "aten::size", "aten::stride"):
            continue
        st = [f for f in (ev.stack or []) if ("ist-net_amd" in f or "bench.py" in f)][:3]
        key = (ev.name, " <- ".join(s.split("/")[-1] for s in st))
        seen[key] = seen.get(key, 0) + 1
for (name, where), n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d} x {name:28s} {where}")
print("---- device activities that are not this library's kernels ----")
dv = {}
for ev in prof.events():
    if ev.device_type.name != "CPU" and not ev.name.startswith("(anonymous namespace)") and "anonymous namespace" not in ev.name.split("(")[0]:
        dv[ev.name[:100]] = dv.get(ev.name[:100], 0) + 1
for k, n in sorted(dv.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d} x {k}")
