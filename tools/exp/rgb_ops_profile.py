"""Which framework ops of the RGB branch's training step launch the copy / elementwise kernels (torch profiler, CPU+CUDA)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from istnet_amd.rgb_branch import ModifiedResnet
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = ModifiedResnet().to(dev).train().to(memory_format=torch.channels_last)
x = torch.randn(32, 3, 192, 192, device=dev).contiguous(memory_format=torch.channels_last)


def step():
    net.zero_grad(set_to_none=True)
    net(x).square().mean().backward()


for _ in range(4):
    step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.device_time_total > 150]
rows.sort(key=lambda e: -e.self_device_time_total)
print(f"{'self_us':>9} {'calls':>5}  op  shapes")
for e in rows[:45]:
    if e.self_device_time_total < 100:
        continue
    print(f"{e.self_device_time_total:9.0f} {e.count:5d}  {e.key[:44]:44s} {str(e.input_shapes)[:110]}")
