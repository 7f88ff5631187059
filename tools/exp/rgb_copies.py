"""Which framework ops launch copy / elementwise kernels in the RGB branch training step (shapes included)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from istnet_amd.rgb_branch import ModifiedResnet
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = ModifiedResnet().to(dev).train().to(memory_format=torch.channels_last)
x = torch.randn(32, 3, 192, 192, device=dev).contiguous(memory_format=torch.channels_last)
def step():
    net.zero_grad(set_to_none=True)
    net(x).square().mean().backward()
for _ in range(4): step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA],
                            record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.self_device_time_total > 30]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:40]:
    print(f"{e.self_device_time_total:9.1f} us x{e.count:3d}  {e.key[:34]:34s} {str(e.input_shapes)[:120]}")
