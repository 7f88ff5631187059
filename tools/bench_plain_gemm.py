"""Plain GEMMs of the step (Z = W0f.feat, FP Zk = Wa.K, level dgrad): own MFMA kernel vs torch.matmul (rocBLAS / hipBLASLt)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, istnet_amd
from istnet_amd import _native
lib = _native.lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
B = 32
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, cin, cout, P in [("SA2 Z", 64, 32, 512), ("SA3 Z", 128, 64, 256), ("SA4 Z", 256, 128, 128), ("FP4 Zk", 512, 512, 64),
                           ("FP3 Zk", 512, 256, 128), ("FP2 Zk", 256, 256, 256), ("FP1 Zk", 256, 128, 512), ("FP4 l1", 512, 512, 128)]:
    x = torch.randn(B, cin, P, device=dev); w = torch.randn(cout, cin, device=dev) * 0.1
    y = torch.empty(B, cout, P, device=dev)
    f = lambda: lib.istnet_pw_forward(B, cin, cout, P, x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), None, None, st)
    g = lambda: torch.matmul(w, x, out=y)
    tf, tg = timeit(f), timeit(g)
    print(f"{name:8s} {cin:4d}>{cout:<4d} n={P:4d}: own {tf:6.1f} us   torch.matmul {tg:6.1f} us")
