"""Timeline of one HIP-graph replay of the bench step from in-stream markers (istnet_debug_marker): when each
stream reaches the end of every level, forward and backward.     python tools/step_timeline.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from istnet_amd import _native
from istnet_amd.optim import FlatAdam, layout_hints

dev = torch.device("cuda:0")
model = bench.make_model(dev)
from istnet_amd.modules import GeometrySlot
batches = [bench.shell_cloud(32, 1024, 0, dev), bench.shell_cloud(32, 1024, 1000, dev)]
slots = [model.prefetch_geometry(bt, GeometrySlot()) for bt in batches]
pts = batches[0]
opt = FlatAdam(model.parameters(), lr=1e-4, adjacent=layout_hints(model))
buf = torch.zeros(256, dtype=torch.int64, device=dev)


def step():
    _native.mark("step start")
    opt.zero_grad(set_to_none=True)
    model.prefetch_geometry(batches[1], slots[1])
    with torch.cuda.stream(__import__("istnet_amd").modules._geometry_stream(dev)):
        _native.mark("geometry of the next batch done (geometry stream)")
    out = model(pts, geometry=slots[0])
    loss, grad = bench.mse_value_and_grad(out)
    _native.mark("loss fwd done")
    out.backward(grad)
    _native.mark("backward joined")
    model.join_geometry()
    opt.step()
    _native.mark("optimizer done")


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
_native.MARKERS = {"buf": buf, "names": []}
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    step()
names = list(_native.MARKERS["names"])
_native.MARKERS = None
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
t = buf[:len(names)].cpu().tolist()
t0 = min(t)
rows = sorted(zip(t, names))
prev = t0
print(f"{'t (us)':>9} {'+delta':>8}  marker   (each marker is itself a ~3 us launch)")
for ti, n in rows:
    print(f"{(ti - t0) / 100.0:9.1f} {(ti - prev) / 100.0:8.1f}  {n}")
    prev = ti
