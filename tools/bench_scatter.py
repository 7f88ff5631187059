"""pw_scatter_dy_kernel in isolation on the encoder's SA2-SA4 layer-0 shapes (B=32): time and algorithmic GB/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, istnet_amd
from istnet_amd import _native
lib = _native.lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
B = 32
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = torch.Generator().manual_seed(0)
for name, n, npoint, s, cout in [("SA2-s16", 512, 256, 16, 32), ("SA2-s32", 512, 256, 32, 32), ("SA3-s16", 256, 128, 16, 64),
                                 ("SA3-s32", 256, 128, 32, 64), ("SA4-s16", 128, 64, 16, 128), ("SA4-s32", 128, 64, 32, 128)]:
    p = npoint * s
    import bench
    from istnet_amd.pointnet2 import _ext as E
    xyz = bench.shell_cloud(B, n, seed=n).to(dev)      # real ball-query indices (padding repeats included)
    new_xyz = E.furthest_point_sampling_gather(xyz, npoint)[1]
    radius = {512: 0.02, 256: 0.04, 128: 0.08}[n] * (2 if s == 32 else 1)
    idx = E.ball_query(new_xyz, xyz, radius, s)
    y = torch.randn(B, cout, p, generator=g).to(dev); d = torch.randn(B, cout, p, generator=g).to(dev)
    bn = torch.stack([torch.ones(cout), torch.zeros(cout), torch.zeros(cout), torch.ones(cout)]).contiguous().to(dev)
    bw = torch.stack([torch.ones(cout), torch.zeros(cout), torch.zeros(cout)]).contiguous().to(dev)
    out = torch.empty(B, cout, n, device=dev); dwx = torch.empty(B, cout, 3, device=dev)
    f = lambda: lib.istnet_pw_scatter_dy(B, cout, n, p, 0, y.data_ptr(), d.data_ptr(), None, 0, None, bn.data_ptr(), bw.data_ptr(),
                                         idx.data_ptr(), out.data_ptr(), 0, xyz.data_ptr(), new_xyz.data_ptr(), s, dwx.data_ptr(), st)
    from istnet_amd.pointnet2 import _ext
    off, ent = _ext.ball_csr(idx, n)
    ch = lib.istnet_pw_scatter_csr_chunks(n)
    dwx2 = torch.empty(B * ch, cout, 3, device=dev)
    f2 = lambda: lib.istnet_pw_scatter_dy_csr(B, cout, n, p, y.data_ptr(), d.data_ptr(), bn.data_ptr(), bw.data_ptr(), off.data_ptr(),
                                              ent.data_ptr(), out.data_ptr(), 0, xyz.data_ptr(), new_xyz.data_ptr(), s, dwx2.data_ptr(), st)
    per_nt = []
    for nt in (512, 1024, 256, 0):           # istnet_pw_set_tuning key 21; the default (0 = by size) last, so it is what stays set
        lib.istnet_pw_set_tuning(21, nt)
        per_nt.append((nt, timeit(f2)))
    t2 = per_nt[-1][1]
    tb = timeit(lambda: _ext.ball_csr(idx, n))
    t = timeit(f)
    nbytes = 4.0 * B * (2 * cout * p + p + cout * n)
    print(f"{name}: cout {cout:4d} P {p:6d}  {t:7.1f} us  {nbytes / t / 1e3:7.0f} GB/s  ({B * ((cout + 3) // 4)} workgroups)   csr: {t2:6.1f} us {nbytes / t2 / 1e3:6.0f} GB/s  (+ list build {tb:5.1f} us, geometry stream)   threads: " + "  ".join(f"{nt}: {t_:.1f} us" for nt, t_ in per_nt))
