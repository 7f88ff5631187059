"""Micro-benchmark of the fused per-point MLP GEMM kernels on the encoder's real layer shapes (B=32).
Prints per layer: time, algorithmic GB/s and TFLOP/s for forward / dgrad / wgrad."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import istnet_amd
from istnet_amd import _native
lib = _native.lib()
dev = torch.device("cuda:0")
B = 32
LAYERS = []  # (name, N per cloud = P, nsample, [cin, c1, c2, ...])
for lvl, (npoint, spec) in enumerate([(512, [3, 16, 16, 32]), (256, [67, 32, 32, 64]), (128, [131, 64, 64, 128]), (64, [259, 128, 128, 256])]):
    for s in (16, 32):
        LAYERS.append((f"SA{lvl+1}-s{s}", npoint * s, s, spec))
for name, n, spec in [("FP3", 128, [768, 512, 512]), ("FP2", 256, [640, 256, 256]), ("FP1", 512, [320, 256, 256]), ("FP0", 1024, [256, 128, 128])]:
    LAYERS.append((name, n, 1, spec))
st = torch.cuda.current_stream().cuda_stream
for kv in os.environ.get("PW_TUNE", "").split(","):   # e.g. PW_TUNE=0:16384,1:512
    if kv:
        k, v = kv.split(":"); assert lib.istnet_pw_set_tuning(int(k), int(v)) == 0

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us

tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
print(f"{'layer':10s} {'cin>cout':>9s} {'N':>7s} | {'fwd us':>7s} {'GB/s':>6s} {'TF/s':>6s} | {'dgrad us':>8s} {'GB/s':>6s} {'TF/s':>6s} | {'wgrad us':>8s} {'GB/s':>6s} {'TF/s':>6s} {'+red us':>7s}")
for name, P, s, spec in LAYERS:
    for li in range(len(spec) - 1):
        cin, cout = spec[li], spec[li + 1]
        x = torch.randn(B, cin, P, device=dev)
        w = torch.randn(cout, cin, device=dev) * 0.1
        wt = w.t().contiguous()
        y = torch.empty(B, cout, P, device=dev)
        nt = max(lib.istnet_pw_stat_tiles(B, cout, P), lib.istnet_pw_forward_tiles(B, cin, cout, P))
        part = torch.empty(2, cout, nt, device=dev)
        insc = torch.rand(cin, device=dev) + 0.5; insh = torch.randn(cin, device=dev) * 0.1
        has_bn = li > 0
        f = lambda: lib.istnet_pw_forward(B, cin, cout, P, x.data_ptr(), w.data_ptr(), insc.data_ptr() if has_bn else None,
                                          insh.data_ptr() if has_bn else None, y.data_ptr(), part[0].data_ptr(), part[1].data_ptr(), st)
        t_f = timeit(f)
        bn = torch.stack([torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1, torch.zeros(cout, device=dev), torch.ones(cout, device=dev)]).contiguous()
        bwdc = torch.stack([torch.ones(cout, device=dev), torch.zeros(cout, device=dev) + 0.01, torch.zeros(cout, device=dev) - 0.01]).contiguous()
        dA = torch.randn(B, cout, P, device=dev)
        dx = torch.empty(B, cin, P, device=dev)
        d = lambda: lib.istnet_pw_dgrad(B, cin, 0, cin, cout, P, 0, w.data_ptr(), y.data_ptr(), dA.data_ptr(), None, 0, None, bn.data_ptr(), bwdc.data_ptr(), dx.data_ptr(), None, None, None, None, st)
        t_d = timeit(d)
        splits = lib.istnet_pw_wgrad_splits(B, cin, cout, P)
        ws = torch.empty(splits, cout, cin, device=dev); dw = torch.empty(cout, cin, device=dev)
        g = lambda: lib.istnet_pw_wgrad(B, cin, cout, P, 0, x.data_ptr(), insc.data_ptr() if has_bn else None, insh.data_ptr() if has_bn else None,
                                        y.data_ptr(), dA.data_ptr(), None, 0, None, bn.data_ptr(), bwdc.data_ptr(), ws.data_ptr(), st)
        t_w = timeit(g)
        r = lambda: lib.istnet_pw_wgrad_reduce(cout * cin, splits, ws.data_ptr(), dw.data_ptr(), st)
        t_r = timeit(r)
        N = B * P
        fl = 2.0 * N * cin * cout
        by_f = 4.0 * N * (cin + cout); by_d = 4.0 * N * (cin + 2 * cout); by_w = 4.0 * N * (cin + 2 * cout)
        tot["fwd"] += t_f; tot["dgrad"] += t_d; tot["wgrad"] += t_w + t_r
        print(f"{name:10s} {cin:4d}>{cout:<4d} {N:7d} | {t_f:7.1f} {by_f/t_f/1e3:6.0f} {fl/t_f/1e6:6.1f} | {t_d:8.1f} {by_d/t_d/1e3:6.0f} {fl/t_d/1e6:6.1f} | {t_w:8.1f} {by_w/t_w/1e3:6.0f} {fl/t_w/1e6:6.1f} {t_r:7.1f}")
print("totals us:", {k: round(v, 1) for k, v in tot.items()})
