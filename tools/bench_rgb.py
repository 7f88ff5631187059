"""RGB branch (ResNet-18/PSP, MIOpen) fwd+bwd at B=32 192x192: memory format / MIOpen find-mode variants."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from istnet_amd.rgb_branch import ModifiedResnet
dev = torch.device("cuda:0")
x = torch.randn(32, 3, 192, 192, device=dev)


def run(tag, channels_last=False, benchmark=False, amp=None):
    torch.backends.cudnn.benchmark = benchmark
    torch.manual_seed(0)
    net = ModifiedResnet().to(dev).train()
    xi = x
    if channels_last:
        net = net.to(memory_format=torch.channels_last)
        xi = x.contiguous(memory_format=torch.channels_last)

    def step():
        net.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=amp) if amp is not None else torch.enable_grad():
            out = net(xi)
        out.float().square().mean().backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    print(f"{tag:40s} {(time.perf_counter() - t0) / 5 * 1e3:8.2f} ms", flush=True)


run("nchw")
run("nchw + benchmark", benchmark=True)
run("channels_last", channels_last=True)
run("channels_last + benchmark", channels_last=True, benchmark=True)
run("nchw bf16 autocast + benchmark", benchmark=True, amp=torch.bfloat16)
