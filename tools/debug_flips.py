"""Count max-pool argmax disagreements (fused vs torch composition) in the B=2 golden encoder."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import istnet_amd
from istnet_amd.modules import PointNet2MSG
from istnet_amd.pointnet2 import fused_mlp, pointnet2_modules
DEV = "cuda:0"
z = np.load("tests/golden/encoder_b2.npz")
CAM = [[0.01, 0.02], [0.02, 0.04], [0.04, 0.08], [0.08, 0.16]]
torch.manual_seed(0)
enc = PointNet2MSG([list(r) for r in CAM]).to(DEV).train()
pts = torch.from_numpy(z["pts"]).to(DEV)
orig = fused_mlp.shared_mlp_maxpool
stats = []
def spy(mlp, x):
    sd = {k: v.clone() for k, v in mlp.state_dict().items()}
    out = orig(mlp, x)
    node = out.grad_fn
    arg_f = node.saved_tensors[1] if node is not None else None
    after = {k: v.clone() for k, v in mlp.state_dict().items()}
    mlp.load_state_dict(sd)
    with torch.no_grad():
        act = mlp(x)
        pooled, arg_t = F.max_pool2d(act, kernel_size=[1, act.size(3)], return_indices=True)
    mlp.load_state_dict(after)
    s = x.shape[3]
    if s > 1 and arg_f is not None and arg_f.numel():
        at = (arg_t.squeeze(-1) % s)
        af = arg_f.long()
        va = torch.gather(act, 3, af.unsqueeze(-1)).squeeze(-1)
        vt = pooled.squeeze(-1)
        true_flip = (va != vt) & (vt > 0)
        dis = (at != af)
        xa = torch.gather(x.unsqueeze(1).expand(-1, act.shape[1], -1, -1, -1), 4, af.unsqueeze(2).unsqueeze(-1).expand(-1, -1, x.shape[1], -1, 1)).squeeze(-1)
        xt = torch.gather(x.unsqueeze(1).expand(-1, act.shape[1], -1, -1, -1), 4, at.unsqueeze(2).unsqueeze(-1).expand(-1, -1, x.shape[1], -1, 1)).squeeze(-1)
        same_col = (xa == xt).all(dim=2)
        gap = (vt - va)
        stats.append((tuple(x.shape), int(dis.sum()), int(true_flip.sum()), "dis&same_input_col", int((dis & same_col).sum()),
                      "max gap/val", float((gap[dis] / vt[dis].clamp_min(1e-30)).max()) if dis.any() else 0.0, float((out - vt).abs().max())))
    else:
        stats.append((tuple(x.shape), -1, -1, float((out - pooled.squeeze(-1)).abs().max())))
    return out
pointnet2_modules.shared_mlp_maxpool = spy
out = enc(pts)
for s_ in stats: print(s_)
