"""Where the step time goes (HIP-graph replay, B=32 N=1024 encoder): geometry only, forward only, forward+backward,
full step; and forward truncated after each SA level.   python tools/phase_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
model = bench.make_model(dev)
pts = bench.shell_cloud(32, 1024, 0, dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True, capturable=True)


def graphed(fn, warm=3):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        fn()
    return g


def timeit(g, n=50):
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def geometry():
    with torch.no_grad():
        model._geometry_prepass(pts[..., 0:3].contiguous())
    torch.cuda.current_stream().wait_stream(__import__("istnet_amd").modules._geometry_stream(dev))


def fwd():
    with torch.no_grad():
        return model(pts)


def fwd_train():
    return model(pts)


def fwd_bwd():
    opt.zero_grad(set_to_none=True)
    model(pts).square().mean().backward()


def full():
    opt.zero_grad(set_to_none=True)
    model(pts).square().mean().backward()
    opt.step()


def sa_only(k):
    def f():
        with torch.no_grad():
            xyz, feat = pts[..., 0:3].contiguous(), None
            for sa in model.SA_modules[:k]:
                xyz, feat = sa(xyz, feat)
            return feat
    return f


res = {}
for name, fn in [("geometry prepass alone", geometry), ("forward (no_grad)", fwd), ("forward (train graph)", fwd_train),
                 ("forward+backward", fwd_bwd), ("full step", full)]:
    res[name] = timeit(graphed(fn))
    print(f"{name:28s} {res[name]:7.3f} ms", flush=True)
os.environ["X"] = "1"
for k in (1, 2, 3, 4):
    print(f"SA levels 1..{k} forward, inline geometry  {timeit(graphed(sa_only(k))):7.3f} ms", flush=True)
