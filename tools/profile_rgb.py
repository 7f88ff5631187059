"""Steady-state kernel breakdown of the RGB branch (ResNet-18 trunk + PSP decoder on MIOpen, channels-last, fp32),
forward + backward at B=32, 192x192, by module group and by kernel (torch profiler, after MIOpen's find phase).

    python tools/profile_rgb.py > profiles/r02_rgb_branch_breakdown.txt"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from istnet_amd.rgb_branch import ModifiedResnet  # noqa: E402

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
t0 = time.perf_counter()
for _ in range(5):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
# per top-level part: time of forward / backward with CUDA events via hooks on the children of the extractor
parts = {}
for name, mod in net.model.named_children() if hasattr(net, "model") else net.named_children():
    parts[name] = mod
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in rows)
print(f"# RGB branch fwd+bwd, B=32 3x192x192, fp32 channels-last on MIOpen: {ms:.2f} ms/step wall; sum of kernel time {tot / 1e3:.2f} ms")
print(f"# top-level children: {list(parts)}")
print(f"{'pct':>6} {'calls':>6} {'total_us':>10} {'avg_us':>9}  kernel")
for e in rows[:70]:
    print(f"{100 * e.device_time_total / tot:6.2f} {e.count:6d} {e.device_time_total:10.1f} {e.device_time_total / e.count:9.1f}  {e.key[:150]}")
