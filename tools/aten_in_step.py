"""Which framework (ATen / runtime) kernels are still inside an encoder training step, and which Python line issues
them: one eager step under torch.profiler with stacks; kernels of libistnet_pn2.so are summarised by count only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from istnet_amd.optim import FlatAdam, layout_hints
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
opt = FlatAdam(model.parameters(), lr=1e-4, adjacent=layout_hints(model))
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
print("---- aten ops by source line (key_averages grouped by stack) ----")
skip = {"aten::empty", "aten::view", "aten::as_strided", "aten::empty_strided", "aten::reshape", "aten::slice", "aten::select",
        "aten::transpose", "aten::t", "aten::expand", "aten::unsqueeze", "aten::squeeze", "aten::_unsafe_view", "aten::detach",
        "aten::alias", "aten::permute", "aten::empty_like", "aten::result_type", "aten::view_as", "aten::narrow", "aten::resize_",
        "aten::lift_fresh", "aten::unbind", "aten::item", "aten::_local_scalar_dense", "aten::is_nonzero", "aten::expand_as",
        "aten::stride", "aten::size", "aten::set_", "aten::_reshape_alias", "aten::unflatten", "aten::flatten", "aten::chunk",
        "aten::split", "aten::index_select_backward", "aten::zeros", "aten::ones", "aten::zero_", "aten::to", "aten::_to_copy"}
rows = []
for ev in prof.key_averages(group_by_stack_n=12):
    if not ev.key.startswith("aten::") or ev.key in skip or ev.device_time_total <= 0:
        continue
    frames = [f for f in ev.stack if "/root/repo" in f or "/ist-net_amd/" in f or "bench.py" in f]
    where = frames[0] if frames else (ev.stack[-1] if ev.stack else "?")
    rows.append((ev.count, ev.device_time_total, ev.key, where))
agg = {}
for cnt, t, key, where in rows:
    k = (key, where)
    c0, t0 = agg.get(k, (0, 0.0))
    agg[k] = (c0 + cnt, t0 + t)
for (key, where), (cnt, t) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:70]:
    print(f"{cnt:4d} x {t:9.1f} us  {key:34s} {where}")
