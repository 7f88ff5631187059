"""Which Python lines issue the framework's (ATen) kernels inside a full-model training step: one eager step under
torch.profiler with stacks, ops with device time grouped by the innermost frame inside this repository."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from istnet_amd import tuned_gemm
from istnet_amd.optim import FlatAdam, layout_hints

tuned_gemm.enable()
from istnet_amd.ist_net import point_branch_side_streams
point_branch_side_streams(False)
dev = torch.device("cuda:0")
model = bench.make_istnet(dev)
fwd = [bench.make_istnet_fwd_bwd(model, bench.istnet_batch(32, 1024, 0, dev))]
opt = FlatAdam(model.parameters(), lr=1e-4, adjacent=layout_hints(model))
step = bench.make_eager_step(fwd, opt, 1)
for _ in range(4):
    step()
torch.cuda.synchronize()
cfg = torch._C._profiler._ExperimentalConfig(verbose=True)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, experimental_config=cfg) as prof:
    step()
    torch.cuda.synchronize()
want = {"aten::copy_", "aten::add_", "aten::add", "aten::mul", "aten::sum", "aten::clone", "aten::contiguous", "aten::fill_",
        "aten::mean", "aten::sub", "aten::div", "aten::cat", "aten::slice_backward", "aten::zeros_like", "aten::mul_",
        "aten::neg", "aten::where", "aten::gather", "aten::bernoulli_", "aten::div_", "aten::index_select", "aten::zero_"}
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.key_averages(group_by_stack_n=30):
    if ev.key not in want or ev.device_time_total <= 0:
        continue
    frames = [f for f in ev.stack if ("ist-net_amd/" in f or "istnet_amd/" in f or "bench.py" in f or "torch/autograd" in f)]
    ours = [f for f in frames if "torch/" not in f]
    inner = [f for f in ours if "bench.py" not in f]
    where = inner[0] if inner else (ours[0] if ours else (frames[0] if frames else (ev.stack[0] if ev.stack else "?")))
    if len(inner) > 1:
        where = inner[0].strip()[-70:] + "  <-  " + inner[1].strip()[-50:]
    k = (ev.key, where.strip()[-130:])
    agg[k][0] += ev.count
    agg[k][1] += ev.device_time_total
print("  n     us   op                where")
for (key, where), (cnt, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{cnt:4d} {t:7.1f}  {key:18s} {where}")
