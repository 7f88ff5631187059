"""Point branch of the full model (BASELINE configs[2] shapes: B=32, N=1024, training): cam + world encoders, IST head,
three pose heads, SupervisedLoss, forward + backward, with the per-point RGB features given -- the concat-free head
stacks (fused_mlp.USE_CONCAT_FREE_HEADS) against the concatenated inputs, HIP-graph replays on one GPU, and the torch
concat kernels each variant launches (from an autograd profile).

    python tools/bench_point_branch.py > profiles/r02_point_branch_heads.txt"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from istnet_amd.ist_net import IST_Net  # noqa: E402
from istnet_amd.losses import SupervisedLoss  # noqa: E402
from istnet_amd.pointnet2 import fused_mlp  # noqa: E402

dev = torch.device("cuda:0")
b, n = 32, 1024
torch.manual_seed(0)
net = IST_Net().to(dev).train()
batch = bench.istnet_batch(b, n, 0, dev)
g = torch.Generator().manual_seed(3)
inputs = {k: batch[k] for k in ("pts", "category_label", "qo", "choose")}
inputs["rgb_local"] = torch.randn(b, 128, n, generator=g).to(dev)
labels = {k: batch[k] for k in ("rotation_label", "translation_label", "size_label", "qo")}
crit = SupervisedLoss(1.0, 10.0)


def step():
    net.zero_grad(set_to_none=True)
    ep = net(inputs)
    ep.update(labels)
    crit(ep).backward()


def graphed():
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    net.zero_grad(set_to_none=True)
    with torch.cuda.graph(gr):
        step()
    for _ in range(5):
        gr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        gr.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 30 * 1e3


def cat_kernels():
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    n_cat = sum(e.count for e in prof.key_averages() if "CatArrayBatchedCopy" in e.key)
    us = sum(e.device_time_total for e in prof.key_averages() if "CatArrayBatchedCopy" in e.key)
    return n_cat, us


print("# point branch of the full model, training step (fwd + bwd, rgb_local given), B=32 N=1024, HIP-graph replay")
for flag in (False, True):
    fused_mlp.USE_CONCAT_FREE_HEADS = flag
    ms = graphed()
    n_cat, us = cat_kernels()
    print(f"concat-free heads {'on ' if flag else 'off'}: {ms:7.3f} ms/step   torch concat kernels per step: {n_cat:3d} ({us:7.1f} us)")
