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
for (cin_total, ci_off, rows, cout, p) in [(768, 0, 512, 512, 64), (768, 512, 256, 512, 128), (640, 0, 512, 256, 128), (320, 0, 256, 256, 256), (256, 0, 256, 128, 512), (512, 0, 512, 512, 128)]:
    w = torch.randn(cout, cin_total, device=dev) * 0.1
    g = torch.randn(B, cout, p, device=dev)
    ident = torch.stack([torch.zeros(cout, device=dev), torch.ones(cout, device=dev), torch.zeros(cout, device=dev), torch.ones(cout, device=dev)]).contiguous()
    bw = torch.stack([torch.ones(cout, device=dev), torch.zeros(cout, device=dev), torch.zeros(cout, device=dev)]).contiguous()
    dx = torch.empty(B, rows, p, device=dev)
    f = lambda: lib.istnet_pw_dgrad(B, cin_total, ci_off, rows, cout, p, 0, w.data_ptr(), g.data_ptr(), g.data_ptr(), None, 0, None, ident.data_ptr(), bw.data_ptr(), dx.data_ptr(), None, None, None, None, st)
    t = timeit(f)
    fl = 2.0 * B * p * rows * cout
    print(f"dgrad rows {rows} (of {cin_total}, off {ci_off}) K={cout} p={p}: {t:7.1f} us  {fl / t / 1e6:6.1f} TF/s  cfg {lib.istnet_pw_dgrad_tile_cfg(B, rows, p)}")
