"""Is the 2 % "fast mode" of the captured step (profiles/r06_run_to_run.txt) a property of the PROCESS or of the CAPTURE?
One process, the pipelined encoder step captured K times (fresh graphs each time, the old ones dropped), each timed the same way."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from istnet_amd.modules import GeometrySlot
from istnet_amd.optim import FlatAdam, layout_hints

dev = torch.device("cuda:0")
model = bench.make_model(dev, seed=0)
opt = FlatAdam(model.parameters(), lr=1e-4, adjacent=layout_hints(model))
batches = [bench.shell_cloud(32, 1024, seed=s, device=dev) for s in (0, 1000)]
slots = [model.prefetch_geometry(bt, GeometrySlot()) for bt in batches]
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def timed(step, windows=3, steps=40):
    for _ in range(10):
        step()
    out = []
    for _ in range(windows):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / steps * 1e3)
    return out


for k in range(K):
    fb = [bench.make_pipelined_fwd_bwd(model, batches, slots, i) for i in (0, 1)]
    step = bench.make_graphed_step(fb, opt, 1, None)
    w = timed(step)
    print(f"capture {k}: {sorted(w)[1]:.4f} ms  {[round(x, 4) for x in w]}", flush=True)
    del step, fb
    torch.cuda.synchronize()
    if k == K // 2 - 1 and "--new-streams" in sys.argv:
        from istnet_amd.pointnet2 import fused_mlp
        from istnet_amd import modules
        fused_mlp._SCALE_STREAMS.clear()
        for name in dir(modules):
            pass
        print("(side-stream pools cleared: new streams from here on)")
