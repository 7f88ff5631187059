import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, istnet_amd
from istnet_amd.pointnet2 import _ext
dev = torch.device("cuda:0"); B = 32
g = torch.Generator().manual_seed(0)
d = torch.randn(B, 1024, 3, generator=g); xyz = (d / d.norm(dim=2, keepdim=True) * 0.1).to(dev).contiguous()
n, npoint, ns, C, r = 128, 64, 32, 256, 0.16
pts = xyz[:, :n].contiguous(); fps = _ext.furthest_point_sampling(pts, npoint)
new = torch.gather(pts, 1, fps.long().unsqueeze(-1).expand(B, npoint, 3)).contiguous()
idx = _ext.ball_query(new, pts, r, ns); go = torch.randn(B, C, npoint, ns, device=dev)
for _ in range(3): _ext.group_points_grad(go, idx, n)
unk = xyz[:, :1024].contiguous(); kn = xyz[:, :512].contiguous()
d2, idx3 = _ext.three_nn(unk, kn); w = torch.rand(B, 1024, 3, device=dev); go2 = torch.randn(B, 256, 1024, device=dev)
for _ in range(3): _ext.three_interpolate_grad(go2, idx3, w, 512)
torch.cuda.synchronize()
