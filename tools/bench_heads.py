"""Forward GEMM rate on the IST / pose-head shapes (B=64, N=2048: M = 131072 points)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, istnet_amd
from istnet_amd import _native
lib = _native.lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
B, P = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 2048)
for kv in os.environ.get("PW_TUNE", "").split(","):
    if kv:
        k, v = kv.split(":"); assert lib.istnet_pw_set_tuning(int(k), int(v)) == 0
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for cin, cout in [(320, 384), (384, 256), (512, 384), (512, 512), (256, 128), (128, 256), (3, 32), (32, 64), (512, 256)]:
    x = torch.randn(B, cin, P, device=dev); w = torch.randn(cout, cin, device=dev) * 0.05
    y = torch.empty(B, cout, P, device=dev)
    sc = torch.ones(cin, device=dev); sh = torch.zeros(cin, device=dev)
    f = lambda: lib.istnet_pw_forward(B, cin, cout, P, x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), y.data_ptr(), None, None, st)
    t = timeit(f)
    fl = 2.0 * B * P * cin * cout
    print(f"{cin:4d}>{cout:<4d} M={B * P}: {t:8.1f} us  {fl / t / 1e6:6.1f} TF/s  {4.0 * B * P * (cin + cout) / t / 1e3:6.0f} GB/s  cfg {lib.istnet_pw_tile_cfg(B, cout, P)}")
