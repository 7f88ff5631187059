import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, istnet_amd
from istnet_amd import _native
lib = _native.lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
B = 32
for (cin, cout, P) in [(128, 128, 2048), (128, 256, 2048), (64, 128, 4096), (512, 512, 128), (256, 256, 512)]:
    x = torch.randn(B, cin, P, device=dev); w = torch.randn(cout, cin, device=dev) * 0.1; wt = w.t().contiguous()
    y = torch.empty(B, cout, P, device=dev); nt = max(lib.istnet_pw_stat_tiles(B, cout, P), lib.istnet_pw_forward_tiles(B, cin, cout, P)); part = torch.empty(2, cout, nt, device=dev)
    bn = torch.stack([torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1, torch.zeros(cout, device=dev), torch.ones(cout, device=dev)]).contiguous()
    bwdc = torch.stack([torch.ones(cout, device=dev), torch.zeros(cout, device=dev) + 0.01, torch.zeros(cout, device=dev) - 0.01]).contiguous()
    dA = torch.randn(B, cout, P, device=dev); dx = torch.empty(B, cin, P, device=dev)
    splits = lib.istnet_pw_wgrad_splits(B, cin, cout, P); ws = torch.empty(splits, cout, cin, device=dev)
    mid = lib.istnet_pw_bwd_mid_ok(cin, cout, P)
    if mid:
        fs = lib.istnet_pw_bwd_mid_splits(B, cin, cout, P); part2 = torch.empty(2, cin, fs, device=dev); ws2 = torch.empty(fs, cout, cin, device=dev)
        bn_in = torch.stack([torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1, torch.zeros(cin, device=dev), torch.ones(cin, device=dev)]).contiguous()
    for _ in range(3):
        lib.istnet_pw_forward(B, cin, cout, P, x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), part[0].data_ptr(), part[1].data_ptr(), st)
        lib.istnet_pw_dgrad(B, cin, 0, cin, cout, P, 0, w.data_ptr(), y.data_ptr(), dA.data_ptr(), None, 0, None, bn.data_ptr(), bwdc.data_ptr(), dx.data_ptr(), None, None, None, None, st)
        lib.istnet_pw_wgrad(B, cin, cout, P, 0, x.data_ptr(), None, None, y.data_ptr(), dA.data_ptr(), None, 0, None, bn.data_ptr(), bwdc.data_ptr(), ws.data_ptr(), st)
        if mid:
            lib.istnet_pw_bwd_mid(B, cin, cout, P, 0, w.data_ptr(), x.data_ptr(), bn_in.data_ptr(), y.data_ptr(), dA.data_ptr(), None, 0, None, bn.data_ptr(), bwdc.data_ptr(), dx.data_ptr(), part2[0].data_ptr(), part2[1].data_ptr(), ws2.data_ptr(), st)
    torch.cuda.synchronize()
