"""The two fused "statistics + finalize" kernels at the head of the backward chains, on the encoder's shapes (B = 32): working-tree
library against ab_base/base.so (tools/build_base.sh <rev>), HIP events, operands re-written by another kernel before every launch so
that they come from HBM / the other XCDs' L2 like in the step.  Checks that both libraries give identical bits."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import istnet_amd  # noqa: F401
from istnet_amd import _native
new = _native.lib()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
base = ctypes.CDLL(os.path.join(ROOT, "ab_base", "base.so")) if os.path.exists(os.path.join(ROOT, "ab_base", "base.so")) else None
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B = 32
i32, f64, i64, vp = ctypes.c_int, ctypes.c_double, ctypes.c_longlong, ctypes.c_void_p
if base is not None:
    base.istnet_bn_bwd_dense_finalize.argtypes = [i32, i32, i32, f64, i32] + [vp] * 8
    base.istnet_bn_bwd_pooled_finalize.argtypes = [i32, i32, i32, f64, i32, vp, i64] + [vp] * 7


def timeit(fn, touch, reps=30):
    for _ in range(3):
        touch(); fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        touch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps * 1e3


print("dense (FP levels):  C, points per cloud -> us base / new")
for c, p in ((128, 1024), (256, 512), (256, 256), (512, 128)):
    y = torch.randn(B, c, p, device=dev); dA = torch.randn(B, c, p, device=dev)
    gamma = torch.rand(c, device=dev) + 0.5
    bn = torch.stack([torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1, torch.randn(c, device=dev) * 0.1, torch.rand(c, device=dev) + 0.5]).contiguous()
    outs = {}
    def touch():
        y.mul_(1.0); dA.mul_(1.0)
    row = []
    for name, lib in (("base", base), ("new", new)):
        if lib is None:
            row.append("   n/a"); continue
        dg, db, bw = torch.empty(c, device=dev), torch.empty(c, device=dev), torch.empty(3, c, device=dev)
        f = lambda: lib.istnet_bn_bwd_dense_finalize(B, c, p, float(B * p), 1, y.data_ptr(), dA.data_ptr(), gamma.data_ptr(), bn.data_ptr(), dg.data_ptr(), db.data_ptr(), bw.data_ptr(), st)
        assert f() == 0
        torch.cuda.synchronize()
        outs[name] = (dg.clone(), db.clone(), bw.clone())
        row.append("%6.1f" % timeit(f, touch))
    same = base is None or all(torch.equal(a, b) for a, b in zip(outs["base"], outs["new"]))
    print(f"  C={c:4d} P={p:5d}: {row[0]} / {row[1]}   {'identical bits' if same else 'BITS DIFFER'}")
print("pooled (SA levels, last layer of a scale):  C, groups per cloud -> us base / new")
for c, g in ((32, 512), (64, 256), (128, 128), (256, 64)):
    ymax = torch.randn(B, c, g, device=dev); dp = torch.randn(B, 2 * c, g, device=dev)     # pooled gradient: a channel slice of the level's output gradient
    gamma = torch.rand(c, device=dev) + 0.5
    bn = torch.stack([torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1, torch.randn(c, device=dev) * 0.1, torch.rand(c, device=dev) + 0.5]).contiguous()
    outs = {}
    def touch():
        ymax.mul_(1.0); dp.mul_(1.0)
    row = []
    for name, lib in (("base", base), ("new", new)):
        if lib is None:
            row.append("   n/a"); continue
        dg, db, bw = torch.empty(c, device=dev), torch.empty(c, device=dev), torch.empty(3, c, device=dev)
        f = lambda: lib.istnet_bn_bwd_pooled_finalize(B, c, g, float(B * g * 32), 1, dp.data_ptr(), 2 * c * g, ymax.data_ptr(), gamma.data_ptr(), bn.data_ptr(), dg.data_ptr(), db.data_ptr(), bw.data_ptr(), st)
        assert f() == 0
        torch.cuda.synchronize()
        outs[name] = (dg.clone(), db.clone(), bw.clone())
        row.append("%6.1f" % timeit(f, touch))
    same = base is None or all(torch.equal(a, b) for a, b in zip(outs["base"], outs["new"]))
    print(f"  C={c:4d} G={g:5d}: {row[0]} / {row[1]}   {'identical bits' if same else 'BITS DIFFER'}")
