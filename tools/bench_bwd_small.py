"""Fused small-layer backward (pw_bwd_small) vs dgrad + wgrad_small on the SA1 / SA2 shapes."""
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


for name, cin, cout, P in [("SA1-s16 16>16", 16, 16, 8192), ("SA1-s16 16>32", 16, 32, 8192), ("SA1-s32 16>16", 16, 16, 16384),
                           ("SA1-s32 16>32", 16, 32, 16384), ("SA2-s16 32>32", 32, 32, 4096), ("SA2-s32 32>32", 32, 32, 8192)]:
    x = torch.randn(B, cin, P, device=dev); w = torch.randn(cout, cin, device=dev) * 0.1
    y = torch.randn(B, cout, P, device=dev); dA = torch.randn(B, cout, P, device=dev)
    mk = lambda c: torch.stack([torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1, torch.zeros(c, device=dev), torch.ones(c, device=dev)]).contiguous()
    bn, bn_in = mk(cout), mk(cin)
    bwdc = torch.stack([torch.ones(cout, device=dev), torch.zeros(cout, device=dev) + 0.01, torch.zeros(cout, device=dev) - 0.01]).contiguous()
    dx = torch.empty(B, cin, P, device=dev)
    nt = lib.istnet_pw_dgrad_stat_tiles(B, cin, P); part = torch.empty(2, cin, nt, device=dev)
    splits = lib.istnet_pw_wgrad_splits(B, cin, cout, P); ws = torch.empty(splits, cout, cin, device=dev)
    fsplits = lib.istnet_pw_bwd_small_splits(B, P); part2 = torch.empty(2, cin, fsplits, device=dev); ws2 = torch.empty(fsplits, cout, cin, device=dev)
    d = lambda: lib.istnet_pw_dgrad(B, cin, 0, cin, cout, P, 0, w.data_ptr(), y.data_ptr(), dA.data_ptr(), None, 0, None, bn.data_ptr(), bwdc.data_ptr(), dx.data_ptr(), x.data_ptr(), bn_in.data_ptr(), part[0].data_ptr(), part[1].data_ptr(), st)
    g = lambda: lib.istnet_pw_wgrad(B, cin, cout, P, 0, x.data_ptr(), bn_in[0].data_ptr(), bn_in[1].data_ptr(), y.data_ptr(), dA.data_ptr(), None, 0, None, bn.data_ptr(), bwdc.data_ptr(), ws.data_ptr(), st)
    f = lambda: lib.istnet_pw_bwd_small(B, cin, cout, P, 0, w.data_ptr(), x.data_ptr(), bn_in.data_ptr(), y.data_ptr(), dA.data_ptr(), None, 0, None, bn.data_ptr(), bwdc.data_ptr(), dx.data_ptr(), part2[0].data_ptr(), part2[1].data_ptr(), ws2.data_ptr(), st)
    td, tg, tf = timeit(d), timeit(g), timeit(f)
    byt = 4.0 * B * P * (2 * cout + 2 * cin)
    print(f"{name:16s} dgrad {td:6.1f}  wgrad {tg:6.1f}  sum {td + tg:6.1f} | fused {tf:6.1f} us  ({byt / tf / 1e3:5.0f} GB/s algorithmic)  splits {splits}")
