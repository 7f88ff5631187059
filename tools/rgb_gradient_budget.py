"""Gradient error budget of the RGB branch on the GPU, B = 2 train mode, dropout off: where do the
differences between the native decoder backward and the framework's come from?

For every parameter gradient, relative L2 distance to a FLOAT64 evaluation of the same module (the arithmetic truth) of
    native   this package's path (fused decoder kernels, native backward)
    torch    the framework's composition of the same modules (rgb_branch.USE_FUSED = False)
    rerun    the SAME framework path run twice: its own run-to-run spread (MIOpen picks algorithms per call; atomics)
and the direct distance native <-> torch that tests/test_rgb_ops_gpu.py bounds.

    python tools/rgb_gradient_budget.py > profiles/r04_rgb_gradient_budget.txt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from istnet_amd import rgb_branch  # noqa: E402

DEV = "cuda:0"
SWITCHES = ("USE_FUSED",)


def build(double=False):
    torch.manual_seed(0)
    net = rgb_branch.ModifiedResnet().train()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    net = net.double() if double else net
    net = net.to(DEV)
    return net if double else net.to(memory_format=torch.channels_last)


def grads(net, x, native):
    saved = {k: getattr(rgb_branch, k) for k in SWITCHES if hasattr(rgb_branch, k)}
    try:
        for k in saved:
            setattr(rgb_branch, k, native)
        net.zero_grad(set_to_none=True)
        out = net(x)
        out.square().mean().backward()
        torch.cuda.synchronize()
        return out.detach().double(), {k: p.grad.detach().double().clone() for k, p in net.named_parameters() if p.grad is not None}
    finally:
        for k, v in saved.items():
            setattr(rgb_branch, k, v)


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def compute(hw=64, b=2):
    torch.manual_seed(1)
    x = torch.randn(b, 3, hw, hw, device=DEV)
    xcl = x.contiguous(memory_format=torch.channels_last)
    o64, g64 = grads(build(True), x.double(), False)
    net = build()
    o_n, g_n = grads(net, xcl, True)
    o_t, g_t = grads(net, xcl, False)
    o_r, g_r = grads(net, xcl, False)
    floor = 1e-4 * max(float(g.norm()) for g in g64.values())      # gradients that are round-off only (conv biases before a train-mode BatchNorm)
    rows = []
    for k in g64:
        if float(g64[k].norm()) < floor:
            continue
        rows.append((k, rel(g_n[k], g64[k]), rel(g_t[k], g64[k]), rel(g_r[k], g_t[k]), rel(g_n[k], g_t[k])))
    outs = (rel(o_n, o64), rel(o_t, o64), rel(o_r, o_t), rel(o_n, o_t))
    return rows, outs


if __name__ == "__main__":
    rows, outs = compute()
    print("# RGB branch (ModifiedResnet, B=2, 64x64, train-mode BatchNorm, dropout off): relative L2 error of every parameter gradient")
    print("# native = this package's kernels, torch = the framework's composition of the same modules, f64 = float64 evaluation")
    print(f"# output:  native vs f64 {outs[0]:.2e}   torch vs f64 {outs[1]:.2e}   torch rerun vs torch {outs[2]:.2e}   native vs torch {outs[3]:.2e}")
    print(f"{'native/f64':>11} {'torch/f64':>11} {'rerun':>10} {'native/torch':>13}  parameter")
    for k, a, t, r, d in sorted(rows, key=lambda r: -r[1]):
        print(f"{a:11.2e} {t:11.2e} {r:10.2e} {d:13.2e}  {k}")
    worst_n, worst_t = max(r[1] for r in rows), max(r[2] for r in rows)
    print(f"# worst: native vs f64 {worst_n:.2e}, torch vs f64 {worst_t:.2e}, rerun {max(r[3] for r in rows):.2e}, "
          f"native vs torch {max(r[4] for r in rows):.2e}; median native/f64 {sorted(r[1] for r in rows)[len(rows) // 2]:.2e}, "
          f"median torch/f64 {sorted(r[2] for r in rows)[len(rows) // 2]:.2e}")
