"""Per-parameter gradient comparison: fused HIP MLP path vs torch composition, golden B=2 encoder."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import istnet_amd
from istnet_amd.modules import PointNet2MSG
from istnet_amd.pointnet2 import pointnet2_modules
DEV = "cuda:0"
z = np.load("tests/golden/encoder_b2.npz")
CAM = [[0.01, 0.02], [0.02, 0.04], [0.04, 0.08], [0.08, 0.16]]
pts = torch.from_numpy(z["pts"]).to(DEV)
def composed(mlp, x):
    act = mlp(x)
    return F.max_pool2d(act, kernel_size=[1, act.size(3)]).squeeze(-1)
fused = pointnet2_modules.shared_mlp_maxpool
res = {}
for name, fn in (("fused", fused), ("torch", composed)):
    pointnet2_modules.shared_mlp_maxpool = fn
    torch.manual_seed(0)
    enc = PointNet2MSG([list(r) for r in CAM]).to(DEV).train()
    out = enc(pts); out.square().mean().backward()
    res[name] = (out.detach(), {n: p.grad.clone() for n, p in enc.named_parameters()})
print("out rel", float((res["fused"][0] - res["torch"][0]).abs().max() / res["torch"][0].abs().max()))
gold = z["grad_norms"]
rows = []
for i, (n, g) in enumerate(res["torch"][1].items()):
    f = res["fused"][1][n]
    rows.append((float((f - g).norm() / (g.norm() + 1e-20)), n, float(g.norm()), float(f.norm()), float(gold[i])))
rows.sort(reverse=True)
for r in rows[:14]: print("%.2e %-48s torch %.6e fused %.6e cpu-golden %.6e" % r)
