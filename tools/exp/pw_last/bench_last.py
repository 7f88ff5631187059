"""Last layer of a set-abstraction scale: stored-activation path vs the activation-free path (pw_last.hip), alone on the
chip.  Stack [cin, cin, cout] on (B, cin, G, S) inputs with the encoder's shapes at B = 32; HIP-event timing of the
forward and of forward + backward (one stream: deferred weight gradients off)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from istnet_amd.pointnet2 import fused_mlp
from istnet_amd.pointnet2.pytorch_utils import SharedMLP

dev = torch.device("cuda:0")
fused_mlp.USE_DEFERRED_WGRAD = False
SHAPES = [(32, 64, 256, 16), (32, 64, 256, 32), (64, 128, 128, 16), (64, 128, 128, 32), (128, 256, 64, 16), (128, 256, 64, 32)]


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for cin, cout, g, s in SHAPES:
    torch.manual_seed(0)
    mlp = SharedMLP([cin, cin, cout], bn=True).to(dev).train()
    x = torch.randn(32, cin, g, s, device=dev)
    wgt = torch.randn(32, cout, g, device=dev)
    res = {}
    for kind in ("old", "new"):
        fused_mlp.USE_POOL_EPILOGUE = kind == "new"

        def fwd():
            with torch.no_grad():
                return fused_mlp.shared_mlp_maxpool(mlp, x)

        def fwd_bwd():
            mlp.zero_grad(set_to_none=True)
            out = fused_mlp.shared_mlp_maxpool(mlp, x.requires_grad_(True))
            (out * wgt).sum().backward()

        res[kind] = (timeit(fwd), timeit(fwd_bwd))
    print(f"cin {cin:3d} cout {cout:3d} G {g:3d} S {s:2d}:  forward {res['old'][0]:7.1f} -> {res['new'][0]:7.1f} us   "
          f"fwd+bwd {res['old'][1]:7.1f} -> {res['new'][1]:7.1f} us   (eager launches, host-bound below ~100 us)")
