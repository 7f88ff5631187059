"""include/istnet_conv.h against the framework's convolution (MIOpen) on the 3x3 / 1x1 layers of the RGB trunk at the
training batch (B = 32, 192 x 192 input): forward, backward-data, backward-weights; time, TFLOP/s, max relative error."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, istnet_amd
from istnet_amd import _native
lib = _native.lib(); dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
st = lambda: torch.cuda.current_stream().cuda_stream
LAYERS = [("layer1 3x3", 64, 64, 3, 1, 48), ("layer2.0 3x3 s2", 64, 128, 3, 2, 48), ("layer2 3x3", 128, 128, 3, 1, 24),
          ("layer2 down 1x1 s2", 64, 128, 1, 2, 48), ("layer3.0 3x3", 128, 256, 3, 1, 24), ("layer3 3x3", 256, 256, 3, 1, 24),
          ("layer3 down 1x1", 128, 256, 1, 1, 24), ("layer4.0 3x3", 256, 512, 3, 1, 24), ("layer4 3x3", 512, 512, 3, 1, 24),
          ("layer4 down 1x1", 256, 512, 1, 1, 24)]
g = torch.Generator().manual_seed(0)
for name, cin, cout, k, s, h in LAYERS:
    pad = k // 2
    x = torch.randn(B, cin, h, h, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, generator=g) * 0.05).to(dev).contiguous(memory_format=torch.channels_last)
    oh = (h + 2 * pad - k) // s + 1
    dy = torch.randn(B, cout, oh, oh, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    flop = 2.0 * B * oh * oh * cout * cin * k * k
    # framework
    y_ref = torch.nn.functional.conv2d(x, w, None, s, pad)
    t_f = timeit(lambda: torch.nn.functional.conv2d(x, w, None, s, pad))
    dx_ref, dw_ref = torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1, [True, True, False])[:2]
    t_bd = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False]))
    t_bw = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False]))
    # native
    y = torch.empty_like(y_ref); dx = torch.empty_like(x); dw = torch.empty_like(w)
    args = (B, h, h, cin, cout, k, k, s, pad)
    wsf = torch.empty(max(1, lib.istnet_conv_workspace_floats(0, *args)), device=dev); wsb = torch.empty(max(1, lib.istnet_conv_workspace_floats(1, *args)), device=dev)
    f = lambda: lib.istnet_conv_forward(*args, x.data_ptr(), w.data_ptr(), y.data_ptr(), wsf.data_ptr(), st())
    assert f() == 0
    n_f = timeit(f)
    bd = lambda: lib.istnet_conv_backward_data(*args, dy.data_ptr(), w.data_ptr(), dx.data_ptr(), wsb.data_ptr(), st())
    assert bd() == 0
    n_bd = timeit(bd)
    sp = lib.istnet_conv_wrw_splits(*args)
    part = torch.empty(sp, w.numel(), device=dev)
    bw = lambda: lib.istnet_conv_backward_weights(*args, x.data_ptr(), dy.data_ptr(), part.data_ptr(), dw.data_ptr(), st())
    assert bw() == 0
    n_bw = timeit(bw)
    rel = lambda a, r: float((a - r).abs().max() / r.abs().max())
    print(f"{name:20s} {flop / 1e9:6.1f} GFLOP | fwd  MIOpen {t_f:7.1f} us {flop / t_f / 1e6:6.1f} TF  native {n_f:7.1f} us {flop / n_f / 1e6:6.1f} TF ws {wsf.numel() * 4 / 2**20:5.0f} MB err {rel(y, y_ref):.1e}"
          f" | bwd-data MIOpen {t_bd:7.1f} native {n_bd:7.1f} us {flop / n_bd / 1e6:6.1f} TF err {rel(dx, dx_ref):.1e}"
          f" | bwd-wgt MIOpen {t_bw:7.1f} native {n_bw:7.1f} us {flop / n_bw / 1e6:6.1f} TF ({sp} splits) err {rel(dw, dw_ref):.1e}")
