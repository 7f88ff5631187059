"""Which Python lines issue the device-to-device copies / framework kernels inside a pipelined encoder step (torch profiler with stacks)."""
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
step = bench.make_eager_step(fwd, opt, 1)
for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
for ev in prof.events():
    if ev.device_type.name == "CPU" and ev.name in ("aten::copy_", "aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::cat", "aten::_foreach_add_", "aten::clone", "aten::contiguous"):
        st = [f for f in (ev.stack or []) if "ist-net_amd" in f or "bench.py" in f][:3]
        print(ev.name, [tuple(s) if not isinstance(s, int) else s for s in (ev.input_shapes or [])][:2], "|", " <- ".join(s.split("/")[-1] for s in st))
