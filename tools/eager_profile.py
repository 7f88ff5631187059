"""Eager encoder step (the reference-style caller, utils/solver.py:88-99): ms / step with torch.optim.Adam and FlatAdam,
host enqueue time, and a cProfile of the host side.   python tools/eager_profile.py [--profile] [--no-streams]"""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from istnet_amd.pointnet2 import fused_mlp
from istnet_amd.optim import FlatAdam, layout_hints

dev = torch.device("cuda:0")
pts = bench.shell_cloud(32, 1024, 0, dev)


def timed(step, n=30, warm=8):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return 1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n


def ref_style_step(model, opt):
    def step():
        opt.zero_grad()
        out = model(pts)
        loss = out.square().mean()
        loss.backward()
        opt.step()
    return step


from istnet_amd import graphed
for auto, streams in ((True, True), (False, True), (False, False)):
    graphed.ENABLED = auto
    fused_mlp.USE_SCALE_STREAMS = fused_mlp.USE_DEFERRED_WGRAD = streams
    for name, mk in (("torch.optim.Adam", lambda m: torch.optim.Adam(m.parameters(), lr=1e-4)),
                     ("torch.optim.Adam(fused)", lambda m: torch.optim.Adam(m.parameters(), lr=1e-4, fused=True)),
                     ("FlatAdam", lambda m: FlatAdam(m.parameters(), lr=1e-4, adjacent=layout_hints(m)))):
        model = bench.make_model(dev)
        opt = mk(model)
        host, total = timed(ref_style_step(model, opt))
        print(f"auto_graph={auto} streams={streams} {name:26s} host enqueue {host:.2f} ms/step, total {total:.2f} ms/step", flush=True)

graphed.ENABLED = False
if "--profile" in sys.argv:
    fused_mlp.USE_SCALE_STREAMS = fused_mlp.USE_DEFERRED_WGRAD = True
    model = bench.make_model(dev)
    opt = FlatAdam(model.parameters(), lr=1e-4, adjacent=layout_hints(model))
    step = ref_style_step(model, opt)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        step()
    pr.disable()
    torch.cuda.synchronize()
    for key in ("tottime", "cumulative"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
        print(s.getvalue()[:9000])
