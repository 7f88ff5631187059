"""SA1 scale-1 SharedMLP on the golden encoder input: fused f32 vs torch f32 vs torch f64 (truth)."""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import istnet_amd
from istnet_amd.modules import PointNet2MSG
from istnet_amd.pointnet2 import pointnet2_modules, fused_mlp
DEV = "cuda:0"
z = np.load("tests/golden/encoder_b2.npz")
CAM = [[0.01, 0.02], [0.02, 0.04], [0.04, 0.08], [0.08, 0.16]]
pts = torch.from_numpy(z["pts"]).to(DEV)
torch.manual_seed(0)
enc = PointNet2MSG([list(r) for r in CAM]).to(DEV).train()
cap = []
orig = pointnet2_modules.shared_mlp_maxpool
def spy(mlp, x):
    cap.append((mlp, x.detach().clone(), copy.deepcopy(mlp.state_dict())))
    return orig(mlp, x)
pointnet2_modules.shared_mlp_maxpool = spy
enc(pts)
which = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mlp, x, sd = cap[which]
print("input", tuple(x.shape))
def run(kind):
    m = copy.deepcopy(mlp); m.load_state_dict(sd); m.train(); m.zero_grad()
    xx = x.clone()
    if kind == "f64": m = m.double(); xx = xx.double()
    if kind == "fused": out = fused_mlp.shared_mlp_maxpool(m, xx)
    else:
        act = m(xx); out = F.max_pool2d(act, kernel_size=[1, act.size(3)]).squeeze(-1)
    w = torch.randn(out.shape, generator=torch.Generator().manual_seed(7)).to(DEV).to(out.dtype)
    (out * w).sum().backward()
    return out.detach().double(), {n: p.grad.double() for n, p in m.named_parameters()}
o64, g64 = run("f64"); of, gf = run("fused"); ot, gt = run("torch")
print("out err: fused %.2e torch %.2e" % (float((of - o64).abs().max() / o64.abs().max()), float((ot - o64).abs().max() / o64.abs().max())))
for n in g64:
    e = lambda g: float((g[n] - g64[n]).norm() / g64[n].norm())
    print("%-32s fused %.2e   torch %.2e" % (n, e(gf), e(gt)))
