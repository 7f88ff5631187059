"""Split-precision convolution (istnet_conv_set_tuning(1, 1): three bf16 terms per operand, six bf16 MFMA products, fp32
accumulation) against the exact-fp32 MFMA kernels on the trunk's layers at B = 32: forward and backward-data, time, TFLOP/s,
and the error of BOTH against a float64 convolution (max |diff| / max |ref|, rms ratio in brackets).
Acceptance: split error <= 2 x the fp32 kernel's error on every layer, gain >= 10 %."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch, istnet_amd
from istnet_amd import _native
lib = _native.lib(); dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32


def timeit(fn, reps=20):
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
print(f"# B={B}; err = max|y - y64| / max|y64| (ratio of the rms errors in brackets); gate: split err <= 2 x fp32-MFMA err, gain >= 10 %")
print("# variants of the split kernel: 1 = K chunks of 16, two workgroups per CU (the default); 2 = K chunks of 32")
ok_all, tot = True, {"fwd": [0.0, 0.0], "bwd": [0.0, 0.0]}
for name, cin, cout, k, s, h in LAYERS:
    pad = k // 2
    # activations with a non-zero mean and a spread of magnitudes (post-ReLU-like), weights kaiming-like
    x = (torch.randn(B, cin, h, h, generator=g).abs() * torch.rand(B, cin, 1, 1, generator=g) * 3).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(dev).contiguous(memory_format=torch.channels_last)
    oh = (h + 2 * pad - k) // s + 1
    dy = torch.randn(B, cout, oh, oh, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    flop = 2.0 * B * oh * oh * cout * cin * k * k
    y64 = torch.nn.functional.conv2d(x.double(), w.double(), None, s, pad)
    dx64 = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), s, pad)
    args = (B, h, h, cin, cout, k, k, s, pad)
    for kind in ("fwd", "bwd"):
        if kind == "bwd" and s != 1:
            continue                      # the native backward-data covers stride 1 (stride 2 stays with the framework)
        ref = y64 if kind == "fwd" else dx64
        res = {}
        for mode in (0, 1, 2):
            assert lib.istnet_conv_set_tuning(1, mode) == 0
            out = torch.empty_like(ref, dtype=torch.float32).contiguous(memory_format=torch.channels_last)
            ws = torch.empty(max(4, lib.istnet_conv_workspace_floats(0 if kind == "fwd" else 1, *args)), device=dev)
            if kind == "fwd":
                f = lambda: lib.istnet_conv_forward(*args, x.data_ptr(), w.data_ptr(), out.data_ptr(), ws.data_ptr(), st())
            else:
                f = lambda: lib.istnet_conv_backward_data(*args, dy.data_ptr(), w.data_ptr(), out.data_ptr(), ws.data_ptr(), st())
            assert f() == 0
            t = timeit(f)
            d = (out.double() - ref)
            res[mode] = (t, float(d.abs().max() / ref.abs().max()), float(d.pow(2).mean().sqrt()))
        lib.istnet_conv_set_tuning(1, 0)
        (t0, e0, r0), (t1, e1, r1) = res[0], res[1]
        ok = e1 <= 2 * e0 and r1 <= 2 * r0
        ok_all &= ok
        tot[kind][0] += t0; tot[kind][1] += t1
        print(f"{name:20s} {kind} {flop / 1e9:6.1f} GFLOP | fp32 MFMA {t0:7.1f} us {flop / t0 / 1e6:6.1f} TF err {e0:.1e} | bf16x3 {t1:7.1f} us "
              f"{flop / t1 / 1e6:6.1f} TF err {e1:.1e} | speed-up {t0 / t1:4.2f}x  err ratio {e1 / e0:4.2f} ({r1 / r0:4.2f}) {'ok' if ok else 'GATE FAILED'}"
              f" | K-32 variant: {res[2][0]:6.1f} us")
for kind in ("fwd", "bwd"):
    print(f"# sum over the layers, {kind}: fp32 MFMA {tot[kind][0]:7.1f} us, split {tot[kind][1]:7.1f} us ({tot[kind][0] / tot[kind][1]:4.2f}x)")
print("# accuracy gate", "PASSED" if ok_all else "FAILED")
