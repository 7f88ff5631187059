"""Split-K forward / dgrad kernels on the feature-propagation launches of the B = 32 encoder step: 32 x 128 tiles (TM = 1,
istnet_pw_set_tuning key 24 = 0) against 64 x 128 tiles (TM = 2, the default where the launch keeps >= 256 workgroups).
Same process, alternating, HIP events; also checks that the two forms give identical bits."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import istnet_amd  # noqa: F401
from istnet_amd import _native
lib = _native.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B = 32
# (points per cloud, cin, cout, what)
SHAPES = [(64, 512, 512, "FP4 known"), (128, 256, 512, "FP4 skip"), (128, 512, 512, "FP4 layer 1"),
          (128, 512, 256, "FP3 known"), (256, 128, 256, "FP3 skip"), (256, 256, 256, "FP3 layer 1"),
          (256, 256, 256, "FP2 known"), (512, 64, 256, "FP2 skip"), (512, 256, 256, "FP2 layer 1"),
          (512, 256, 128, "FP1 known"), (1024, 128, 128, "FP1 layer 1")]


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"{'launch':14s} {'rows':>6s} {'cin>cout':>9s} | fwd us TM1 / TM2  (TF/s)        | dgrad us TM1 / TM2  (TF/s)      | kernel fwd / dgrad")
tot = [0.0, 0.0, 0.0, 0.0]
for P, cin, cout, what in SHAPES:
    x = torch.randn(B, cin, P, device=dev)
    w = torch.randn(cout, cin, device=dev) * 0.1
    y = torch.empty(B, cout, P, device=dev)
    nt = lib.istnet_pw_forward_ld_tiles(B, cin, cout, P)
    part = torch.empty(2, cout, max(nt, 1), device=dev)
    insc, insh = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1
    bn = torch.stack([torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1, torch.zeros(cout, device=dev),
                      torch.ones(cout, device=dev)]).contiguous()
    bwdc = torch.stack([torch.ones(cout, device=dev), torch.zeros(cout, device=dev) + 0.01, torch.zeros(cout, device=dev) - 0.01]).contiguous()
    bn_in = torch.stack([insc, insh, torch.zeros(cin, device=dev), torch.ones(cin, device=dev)]).contiguous()
    dA = torch.randn(B, cout, P, device=dev)
    dx = torch.empty(B, cin, P, device=dev)
    ntd = lib.istnet_pw_dgrad_tiles(B, cin, cout, P, 1)
    pg = torch.empty(2, cin, max(ntd, 1), device=dev)
    fwd = lambda: lib.istnet_pw_forward_ld(B, cin, cout, P, x.data_ptr(), w.data_ptr(), cin, insc.data_ptr(), insh.data_ptr(),
                                           y.data_ptr(), part[0].data_ptr(), part[1].data_ptr(), st)
    dgr = lambda: lib.istnet_pw_dgrad(B, cin, 0, cin, cout, P, 0, w.data_ptr(), y.data_ptr(), dA.data_ptr(), None, 0, None,
                                      bn.data_ptr(), bwdc.data_ptr(), dx.data_ptr(), x.data_ptr(), bn_in.data_ptr(),
                                      pg[0].data_ptr(), pg[1].data_ptr(), st)
    res, outs = {}, {}
    for tm_key in (0, 256, 0, 256):
        lib.istnet_pw_set_tuning(24, tm_key)
        assert fwd() == 0 and dgr() == 0
        torch.cuda.synchronize()
        outs.setdefault(tm_key, (y.clone(), part.clone(), dx.clone(), pg.clone()))
        res.setdefault(tm_key, []).append((timeit(fwd), timeit(dgr)))
    lib.istnet_pw_set_tuning(24, 256)
    same = all(torch.equal(a, b) for a, b in zip(outs[0], outs[256]))
    f1, d1 = (min(r[i] for r in res[0]) for i in (0, 1))
    f2, d2 = (min(r[i] for r in res[256]) for i in (0, 1))
    fl = 2.0 * B * P * cin * cout
    ks = "fwd sk<%d>" % lib.istnet_pw_sk_tm(B, cout, P) if lib.istnet_pw_forward_cfg(B, cin, cout, P) == 1 else "fwd other"
    kd = "dgrad sk<%d>" % lib.istnet_pw_sk_tm(B, cin, P) if lib.istnet_pw_dgrad_sk(B, cin, cout, P) else "dgrad other"
    tot[0] += f1; tot[1] += f2; tot[2] += d1; tot[3] += d2
    print(f"{what:14s} {B * P:6d} {cin:4d}>{cout:<4d} | {f1:6.1f} / {f2:6.1f}  ({fl / f1 / 1e6:5.1f} / {fl / f2 / 1e6:5.1f}) | "
          f"{d1:6.1f} / {d2:6.1f}  ({fl / d1 / 1e6:5.1f} / {fl / d2 / 1e6:5.1f}) | {ks}, {kd}{'' if same else '   BITS DIFFER'}")
print("totals us: fwd TM1 %.1f TM2 %.1f | dgrad TM1 %.1f TM2 %.1f" % tuple(tot))
