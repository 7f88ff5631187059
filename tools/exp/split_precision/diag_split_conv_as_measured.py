"""Timing experiments on the 8-wave split kernel (wrong results on purpose): which part of a chunk costs what."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch, istnet_amd
from istnet_amd import _native
lib = _native.lib(); dev = torch.device("cuda:0")
B, cin, cout, k, s, h = 32, 512, 512, 3, 1, 24
x = torch.randn(B, cin, h, h).to(dev).contiguous(memory_format=torch.channels_last)
w = (torch.randn(cout, cin, k, k) * 0.02).to(dev).contiguous(memory_format=torch.channels_last)
y = torch.empty((B, cout, h, h), device=dev).contiguous(memory_format=torch.channels_last)
args = (B, h, h, cin, cout, k, k, s, 1)
st = lambda: torch.cuda.current_stream().cuda_stream
names = {0: "fp32 MFMA kernel", 2: "split, 8 waves (full)", 4: "  without the MFMAs", 5: "  without split + LDS writes", 6: "  without LDS reads",
         7: "  without global loads", 1: "split W variant", 3: "split, 4 waves"}
for mode in (0, 2, 4, 5, 6, 7, 1, 3):
    lib.istnet_conv_set_tuning(1, mode)
    ws = torch.empty(max(1, lib.istnet_conv_workspace_floats(0, *args)), device=dev)
    f = lambda: lib.istnet_conv_forward(*args, x.data_ptr(), w.data_ptr(), y.data_ptr(), ws.data_ptr(), st())
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{names[mode]:32s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us")
lib.istnet_conv_set_tuning(1, 0)
