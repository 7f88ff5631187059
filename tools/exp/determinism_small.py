"""Is the plain (launch-by-launch) encoder step bit-reproducible on the small odd shape of tests/test_autograph_gpu.py
(B = 2, N = 256: level 1 samples 512 of 256 points)?  Repeats the same step and lists parameters whose gradient bits differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import istnet_amd
from istnet_amd import graphed
from istnet_amd.modules import PointNet2MSG
CAM = [[0.01, 0.02], [0.02, 0.04], [0.04, 0.08], [0.08, 0.16]]
graphed.ENABLED = False


def cloud(b, n, seed):
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(b, n, 3, generator=g)
    return (d / d.norm(dim=2, keepdim=True) * 0.1 + torch.randn(b, n, 3, generator=g) * 0.002).cuda().contiguous()


for b, n in ((2, 256), (2, 512), (4, 1024)):
    torch.manual_seed(0)
    model = PointNet2MSG([list(r) for r in CAM]).cuda().train()
    names = [k for k, _ in model.named_parameters()]
    a, c = cloud(b, n, 1), cloud(b, n, 2)
    ref = None
    bad = {}
    for it in range(40):
        model.zero_grad()
        junk = torch.empty(1 << (10 + it % 12), device="cuda").normal_()      # move the allocator around
        (model(a).square().mean() + model(c).square().mean()).backward()
        got = [p.grad.clone() for p in model.parameters()]
        if ref is None:
            ref = got
        else:
            for k, g, r in zip(names, got, ref):
                if not torch.equal(g, r):
                    bad[k] = bad.get(k, 0) + 1
        del junk
    print(f"B={b} N={n}: parameters whose gradient differed from run 0 in 39 repeats:", bad or "none")
