import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import istnet_amd
from istnet_amd.pointnet2.pytorch_utils import SharedMLP
from istnet_amd.pointnet2.fused_mlp import shared_mlp_maxpool
DEV = "cuda:0"
spec, b, g, s = [3, 16, 16, 32], 2, 512, 32
if len(sys.argv) > 1:
    spec = eval(sys.argv[1]); b, g, s = map(int, sys.argv[2:5])
dup = len(sys.argv) <= 5 or sys.argv[5] == "1"
torch.manual_seed(sum(spec) + g)
A = SharedMLP(list(spec), bn=True).to(DEV); B = SharedMLP(list(spec), bn=True).to(DEV); B.load_state_dict(A.state_dict())
if len(sys.argv) > 6 and sys.argv[6] == "1":
    with torch.no_grad():
        for m in (A, B):
            gen = torch.Generator().manual_seed(3)
            for mod in m.modules():
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.weight.copy_((torch.rand(mod.weight.shape, generator=gen) + 0.5).to(DEV))
                    mod.bias.copy_((torch.randn(mod.bias.shape, generator=gen) * 0.2).to(DEV))
                    mod.running_mean.copy_((torch.randn(mod.bias.shape, generator=gen) * 0.1).to(DEV))
                    mod.running_var.copy_((torch.rand(mod.bias.shape, generator=gen) + 0.5).to(DEV))
x = (torch.randn(b, spec[0], g, s, generator=torch.Generator().manual_seed(1)) * 1.5 + 0.3).to(DEV)
if s > 1 and dup:
    x[:, :, :, s // 2:] = x[:, :, :, :1]
def run(m, fused):
    m.train(); m.zero_grad(); xx = x.clone().requires_grad_(True)
    if fused: out = shared_mlp_maxpool(m, xx)
    else:
        act = m(xx); out = F.max_pool2d(act, kernel_size=[1, act.size(3)]).squeeze(-1)
    w = torch.randn(out.shape, generator=torch.Generator().manual_seed(7)).to(DEV)
    (out * w).sum().backward()
    return out.detach(), xx.grad, {n: p.grad for n, p in m.named_parameters()}
of, dxf, gf = run(A, True); ot, dxt, gt = run(B, False)
rel = lambda a, c: float((a - c).abs().max() / (c.abs().max() + 1e-12))
print("out", rel(of, ot))
for n in gt:
    dd = (gf[n] - gt[n]).abs().flatten(); mx = gt[n].abs().max()
    print(n, rel(gf[n], gt[n]), "n>1e-4:", int((dd > 1e-4 * mx).sum()), "of", dd.numel(), "median rel", float(dd.median() / mx))
# activation-level comparison of the forward
acts = {}
xx = x.clone()
B.train()
with torch.no_grad():
    h = xx
    for li, unit in enumerate(B):
        yraw = unit.conv(h); acts[li] = yraw
        h = unit.activation(unit.normlayer(yraw.clone()))
    pool_t, arg_t = F.max_pool2d(h, kernel_size=[1, h.size(3)], return_indices=True)
    # how close are the top-2 values in each group?
    top2 = h.topk(2, dim=3).values
    gap = (top2[..., 0] - top2[..., 1])
    print("groups", gap.numel(), "gap==0:", int((gap == 0).sum()), "gap<1e-6 & max>0:", int(((gap < 1e-6) & (top2[..., 0] > 0)).sum()))
if s > 1 and dup:
    fold = lambda d: torch.cat([d[..., :1] + d[..., s // 2:].sum(-1, keepdim=True), d[..., 1:s // 2]], dim=-1)
    dxf, dxt = fold(dxf), fold(dxt)
d = (dxf - dxt).abs()
print("L1 rel", float(d.sum() / dxt.abs().sum()))
print("dx", rel(dxf, dxt), "num bad", int((d > 1e-4 * dxt.abs().max()).sum()), "of", d.numel())
bad = (d > 1e-4 * dxt.abs().max()).nonzero()
print(bad[:20].tolist())
for i in bad[:6].tolist():
    print(i, float(dxf[tuple(i)]), float(dxt[tuple(i)]))
