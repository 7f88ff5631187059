"""Inference throughput of the IST-Net point branch (BASELINE config 5 shape: eval mode, B=64, N=2048, RGB features
given) -- eager and HIP-graph replay -- plus a kernel count.      python tools/bench_infer.py [B] [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from istnet_amd.ist_net import IST_Net

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = IST_Net().to(dev).eval()
g = torch.Generator().manual_seed(1)
batch = {"pts": (bench.shell_cloud(B, N, 0) + torch.tensor([0.0, 0.0, 0.8])).to(dev),
         "rgb_local": torch.randn(B, 128, N, generator=g).to(dev),
         "category_label": torch.randint(0, 6, (B, 1), generator=g).to(dev)}


def fwd():
    with torch.no_grad():
        return net(batch)


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


ms = timeit(fwd)
print(f"eager   B={B} N={N}: {ms:7.3f} ms/batch  {B / ms * 1e3:9.0f} clouds/s")
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        fwd()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    out = fwd()
ms = timeit(gr.replay)
print(f"hipgraph B={B} N={N}: {ms:7.3f} ms/batch  {B / ms * 1e3:9.0f} clouds/s")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    fwd()
    torch.cuda.synchronize()
evs = [e for e in prof.key_averages() if e.device_time_total > 0]
evs.sort(key=lambda e: -e.device_time_total)
print(f"kernels in one eager forward: {sum(e.count for e in evs)}; top by time:")
for e in evs[:14]:
    print(f"  {e.device_time_total:9.1f} us  x{e.count:4d}  {e.key[:90]}")
