"""How long does a small chain kernel (bn_finalize_bwd, 128 workgroups) take when a big GEMM kernel of another stream
occupies the chip?  Prints solo / co-run durations per big kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, istnet_amd
from istnet_amd import _native
lib = _native.lib(); dev = torch.device("cuda:0")
B, cin, cout, P = 32, 128, 128, 2048
x = torch.randn(B, cin, P, device=dev); w = torch.randn(cout, cin, device=dev) * 0.1
y = torch.randn(B, cout, P, device=dev); dA = torch.randn(B, cout, P, device=dev)
mk = lambda c: torch.stack([torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1, torch.zeros(c, device=dev), torch.ones(c, device=dev)]).contiguous()
bn, bn_in = mk(cout), mk(cin)
bwdc = torch.stack([torch.ones(cout, device=dev), torch.zeros(cout, device=dev) + 0.01, torch.zeros(cout, device=dev) - 0.01]).contiguous()
dx = torch.empty(B, cin, P, device=dev)
fs = lib.istnet_pw_bwd_mid_splits(B, cin, cout, P); part2 = torch.empty(2, cin, fs, device=dev); ws2 = torch.empty(fs, cout, cin, device=dev)
nt = lib.istnet_pw_dgrad_stat_tiles(B, cin, P); part = torch.empty(2, cin, nt, device=dev)
sp = lib.istnet_pw_wgrad_splits(B, cin, cout, P); ws = torch.empty(sp, cout, cin, device=dev)
ntf = lib.istnet_pw_forward_tiles(B, cin, cout, P); partf = torch.empty(2, cout, ntf, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
big = {
    "pw_bwd_mid<4,4>": lambda st: lib.istnet_pw_bwd_mid(B, cin, cout, P, 0, w.data_ptr(), x.data_ptr(), bn_in.data_ptr(), y.data_ptr(), dA.data_ptr(), None, 0, None, bn.data_ptr(), bwdc.data_ptr(), dx.data_ptr(), part2[0].data_ptr(), part2[1].data_ptr(), ws2.data_ptr(), st),
    "pw_dgrad": lambda st: lib.istnet_pw_dgrad(B, cin, 0, cin, cout, P, 0, w.data_ptr(), y.data_ptr(), dA.data_ptr(), None, 0, None, bn.data_ptr(), bwdc.data_ptr(), dx.data_ptr(), x.data_ptr(), bn_in.data_ptr(), part[0].data_ptr(), part[1].data_ptr(), st),
    "pw_wgrad2": lambda st: lib.istnet_pw_wgrad(B, cin, cout, P, 0, x.data_ptr(), bn_in[0].data_ptr(), bn_in[1].data_ptr(), y.data_ptr(), dA.data_ptr(), None, 0, None, bn.data_ptr(), bwdc.data_ptr(), ws.data_ptr(), st),
    "pw_fwd": lambda st: lib.istnet_pw_forward(B, cin, cout, P, x.data_ptr(), w.data_ptr(), bn_in[0].data_ptr(), bn_in[1].data_ptr(), y.data_ptr(), partf[0].data_ptr(), partf[1].data_ptr(), st),
}
gamma = torch.ones(cin, device=dev); dgm = torch.empty(cin, device=dev); dbt = torch.empty(cin, device=dev); bw2 = torch.empty(3, cin, device=dev)
small = lambda st: lib.istnet_bn_finalize_bwd(cin, fs, float(B * P), 1, part2[0].data_ptr(), part2[1].data_ptr(), gamma.data_ptr(), bn_in.data_ptr(), dgm.data_ptr(), dbt.data_ptr(), bw2.data_ptr(), st)
filler = torch.empty(64 << 20, device=dev)


def time_small(big_fn):
    ts = []
    for _ in range(12):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s1):
            if big_fn is not None:
                b0.record(s1)
                for _ in range(3):
                    assert big_fn(s1.cuda_stream) == 0
                b1.record(s1)
        with torch.cuda.stream(s2):
            filler[: 1 << 20].zero_()          # lets the big kernels start first
            e0.record(s2)
            assert small(s2.cuda_stream) == 0
            e1.record(s2)
        torch.cuda.synchronize()
        ts.append((e0.elapsed_time(e1) * 1e3, b0.elapsed_time(b1) * 1e3 / 3 if big_fn is not None else 0.0))
    ts.sort()
    return ts[len(ts) // 2]


print(f"solo: small {time_small(None)[0]:.1f} us")
for name, fn in big.items():
    t, tb = time_small(fn)
    print(f"beside {name:18s} (~{tb:.0f} us each): small kernel takes {t:.1f} us")
