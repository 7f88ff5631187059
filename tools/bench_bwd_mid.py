"""Fused mid-size-layer backward (pw_bwd_mid) vs the dgrad + wgrad pair on the encoder's 64 / 128-channel layers."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, istnet_amd
from istnet_amd import _native
lib = _native.lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
B = 32
for kv in os.environ.get("PW_TUNE", "").split(","):
    if kv:
        k, v = kv.split(":"); assert lib.istnet_pw_set_tuning(int(k), int(v)) == 0


def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tot = [0.0, 0.0]
print("layer              pooled |  dgrad   wgrad     sum |  fused us   TF/s   GB/s algorithmic   splits")
for name, cin, cout, P, S in [("SA2-s16 32>64", 32, 64, 4096, 16), ("SA2-s32 32>64", 32, 64, 8192, 32),
                              ("SA3-s16 64>64", 64, 64, 2048, 0), ("SA3-s16 64>128", 64, 128, 2048, 16),
                              ("SA3-s32 64>64", 64, 64, 4096, 0), ("SA3-s32 64>128", 64, 128, 4096, 32),
                              ("SA4-s16 128>128", 128, 128, 1024, 0), ("SA4-s32 128>128", 128, 128, 2048, 0),
                              ("FP0 128>128", 128, 128, 1024, 0)]:
    x = torch.randn(B, cin, P, device=dev); w = torch.randn(cout, cin, device=dev) * 0.1
    y = torch.randn(B, cout, P, device=dev)
    mk = lambda c: torch.stack([torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1, torch.zeros(c, device=dev), torch.ones(c, device=dev)]).contiguous()
    bn, bn_in = mk(cout), mk(cin)
    bwdc = torch.stack([torch.ones(cout, device=dev), torch.zeros(cout, device=dev) + 0.01, torch.zeros(cout, device=dev) - 0.01]).contiguous()
    if S:
        dpool = torch.randn(B, cout, P // S, device=dev); arg = torch.randint(0, S, (B, cout, P // S), device=dev, dtype=torch.uint8)
        src = (None, dpool.data_ptr(), 0, arg.data_ptr())
        gbytes = B * cout * (P // S) * 5
    else:
        dA = torch.randn(B, cout, P, device=dev)
        src = (dA.data_ptr(), None, 0, None)
        gbytes = 4 * B * cout * P
    dx = torch.empty(B, cin, P, device=dev)
    nt = lib.istnet_pw_dgrad_stat_tiles(B, cin, P); part = torch.empty(2, cin, nt, device=dev)
    splits = lib.istnet_pw_wgrad_splits(B, cin, cout, P); ws = torch.empty(splits, cout, cin, device=dev)
    fsplits = lib.istnet_pw_bwd_mid_splits(B, cin, cout, P); part2 = torch.empty(2, cin, fsplits, device=dev); ws2 = torch.empty(fsplits, cout, cin, device=dev)
    d = lambda: lib.istnet_pw_dgrad(B, cin, 0, cin, cout, P, S, w.data_ptr(), y.data_ptr(), *src, bn.data_ptr(), bwdc.data_ptr(), dx.data_ptr(), x.data_ptr(), bn_in.data_ptr(), part[0].data_ptr(), part[1].data_ptr(), st)
    g = lambda: lib.istnet_pw_wgrad(B, cin, cout, P, S, x.data_ptr(), bn_in[0].data_ptr(), bn_in[1].data_ptr(), y.data_ptr(), *src, bn.data_ptr(), bwdc.data_ptr(), ws.data_ptr(), st)
    f = lambda: lib.istnet_pw_bwd_mid(B, cin, cout, P, S, w.data_ptr(), x.data_ptr(), bn_in.data_ptr(), y.data_ptr(), *src, bn.data_ptr(), bwdc.data_ptr(), dx.data_ptr(), part2[0].data_ptr(), part2[1].data_ptr(), ws2.data_ptr(), st)
    assert d() == 0 and g() == 0 and f() == 0
    td, tg, tf = timeit(d), timeit(g), timeit(f)
    byt = 4.0 * B * P * (cout + 2 * cin) + gbytes + 4.0 * fsplits * cout * cin
    fl = 4.0 * B * P * cin * cout
    tot[0] += td + tg; tot[1] += tf
    print(f"{name:18s} {str(bool(S)):6s} | {td:6.1f}  {tg:6.1f}  {td + tg:6.1f} |  {tf:7.1f}  {fl / tf / 1e6:5.1f}  {byt / tf / 1e3:6.0f}             {fsplits}")
print(f"totals: pair {tot[0]:.1f} us, fused {tot[1]:.1f} us")
