import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from istnet_amd.pointnet2.pytorch_utils import SharedMLP
from istnet_amd.pointnet2 import fused_mlp
DEV = "cuda:0"
for ratio in (0.0, 3.0, 10.0, 30.0):
    torch.manual_seed(2)
    b, c, g, s = 8, 16, 256, 16
    mlp = SharedMLP([c, c, 32], bn=True).to(DEV).train()
    with torch.no_grad():
        mlp[0].conv.weight.copy_(torch.eye(c).view(c, c, 1, 1))
    gen = torch.Generator().manual_seed(3)
    sign = torch.where(torch.arange(c) % 2 == 0, 1.0, -1.0).view(1, c, 1, 1)
    x = (torch.randn(b, c, g, s, generator=gen) + ratio * sign).to(DEV)
    wgt = torch.randn(b, 32, g, generator=gen).to(DEV)
    def run(m, xx, fused_small=True):
        m.zero_grad(set_to_none=True)
        fused_mlp.USE_FUSED_SMALL_BWD = fused_small
        if xx.dtype == torch.float64:
            act = m(xx); out = F.max_pool2d(act, kernel_size=[1, act.size(3)]).squeeze(-1)
        else:
            out = fused_mlp.shared_mlp_maxpool(m, xx)
        (out * wgt.to(out.dtype)).sum().backward()
        return out.detach(), {n: p.grad.detach().clone() for n, p in m.named_parameters()}
    o64, g64 = run(copy.deepcopy(mlp).double(), x.double())
    for fs in (True, False):
        o, gr = run(copy.deepcopy(mlp), x, fs)
        print(f"ratio {ratio} fused_small={fs}: out {float((o.double()-o64).abs().max()/o64.abs().max()):.1e}", end="  ")
        for n in gr:
            d = (gr[n].double() - g64[n]).abs()
            print(f"{n.split('.')[0]}.{n.split('.')[-1]} {float(d.max()/g64[n].abs().max()):.1e}", end=" ")
        print()
