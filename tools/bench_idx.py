"""Micro-benchmark of the index ops on the encoder's shapes (B=32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, istnet_amd
from istnet_amd import _native
from istnet_amd.pointnet2 import _ext
lib = _native.lib(); dev = torch.device("cuda:0")
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
B = 32
g = torch.Generator().manual_seed(0)
d = torch.randn(B, 1024, 3, generator=g); xyz = (d / d.norm(dim=2, keepdim=True) * 0.1).to(dev).contiguous()
for mw in (100000, 512, 256, 128):
    lib.istnet_pn2_set_tuning(0, mw)
    line = f"fps multiwave_min={mw}: "
    cur = xyz
    for n, m in ((1024, 512), (512, 256), (256, 128), (128, 64)):
        pts = xyz[:, :n].contiguous()
        line += f"n={n}:{timeit(lambda: _ext.furthest_point_sampling(pts, m)):.1f}us  "
    print(line)
lib.istnet_pn2_set_tuning(0, 512)
# scatter kernels on real-ish index patterns
for (n, npoint, ns, C, r) in ((512, 256, 32, 64, 0.04), (256, 128, 32, 128, 0.08), (128, 64, 32, 256, 0.16), (128, 64, 16, 256, 0.08)):
    pts = xyz[:, :n].contiguous()
    fps = _ext.furthest_point_sampling(pts, npoint)
    new = torch.gather(pts, 1, fps.long().unsqueeze(-1).expand(B, npoint, 3)).contiguous()
    idx = _ext.ball_query(new, pts, r, ns)
    go = torch.randn(B, C, npoint, ns, device=dev)
    t = timeit(lambda: _ext.group_points_grad(go, idx, n))
    print(f"group_grad C={C} n={n} npoint={npoint} ns={ns}: {t:.1f}us  {go.numel()*4/t/1e3:.0f} GB/s")
for (n, m, C) in ((128, 64, 512), (256, 128, 512), (512, 256, 256), (1024, 512, 256)):
    unk = xyz[:, :n].contiguous(); kn = xyz[:, :m].contiguous()
    d2, idx = _ext.three_nn(unk, kn)
    w = torch.rand(B, n, 3, device=dev)
    go = torch.randn(B, C, n, device=dev)
    t = timeit(lambda: _ext.three_interpolate_grad(go, idx, w, m))
    t2 = timeit(lambda: _ext.three_nn(unk, kn))
    print(f"interp_grad C={C} n={n} m={m}: {t:.1f}us {go.numel()*4/t/1e3:.0f} GB/s   three_nn {t2:.1f}us")
