import sys, os
sys.path.insert(0, "/root/repo")
import torch, istnet_amd
from istnet_amd import _native
lib = _native.lib(); dev = torch.device("cuda:0")
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
st = lambda: torch.cuda.current_stream().cuda_stream
for cin, cout in [(320, 384), (384, 256), (512, 384), (512, 512), (256, 128), (128, 256), (512, 256), (1024, 512)]:
    if cin % 64 or cout % 64: continue
    B, h = 32, 64
    x = torch.randn(B, cin, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 1, 1, device=dev) * 0.05)
    y = torch.empty(B, cout, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    args = (B, h, h, cin, cout, 1, 1, 1, 0)
    ws = torch.empty(max(1, lib.istnet_conv_workspace_floats(0, *args)), device=dev)
    f = lambda: lib.istnet_conv_forward(*args, x.data_ptr(), w.data_ptr(), y.data_ptr(), ws.data_ptr(), st())
    assert f() == 0
    t = timeit(f); fl = 2.0 * B * h * h * cin * cout
    print(f"{cin}>{cout} M={B*h*h}: {t:7.1f} us {fl / t / 1e6:6.1f} TF  ws {ws.numel()*4/2**20:.0f} MB")
