import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from istnet_amd import rgb_branch, _native
dev = torch.device("cuda:0")
rows = 32 * 192 * 192
u = torch.randn(rows, 64, device=dev)
a = torch.randn(64, 64, device=dev) * 0.1
c0 = torch.randn(64, device=dev)
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for native in (True, False):
    rgb_branch.USE_NATIVE_MOMENTS = native
    print("moments native" if native else "moments torch ", round(timeit(lambda: rgb_branch._moments(u)), 1), "us")
out = torch.empty_like(u)
lib = _native.lib()
print("rowmix native", round(timeit(lambda: lib.istnet_nhwc_rowmix64(rows, u.data_ptr(), a.data_ptr(), c0.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)), 1), "us")
print("addmm        ", round(timeit(lambda: torch.addmm(c0, u, a.t())), 1), "us")
