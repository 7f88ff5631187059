"""Timeline of one HIP-graph replay of the full-model (config 3) training step from in-stream markers: when the RGB branch, the
encoders, the heads and their backward passes start and end on their streams.   python tools/istnet_step_timeline.py [--eager]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from istnet_amd import _native, tuned_gemm, ist_net
from istnet_amd.losses import SupervisedLoss
from istnet_amd.optim import FlatAdam, layout_hints

tuned_gemm.enable()
dev = torch.device("cuda:0")
model = bench.make_istnet(dev)
batch = bench.istnet_batch(32, 1024, 0, dev)
opt = FlatAdam(model.parameters(), lr=1e-4, adjacent=layout_hints(model))
crit = SupervisedLoss(1.0, 10.0)
buf = torch.zeros(512, dtype=torch.int64, device=dev)
rgb_stream = ist_net._rgb_stream(dev)
mark = _native.mark


def tap(t, name):
    """Mark when the gradient of ``t`` is produced (on the stream that produces it)."""
    if t.requires_grad:
        t.register_hook(lambda g: (mark(name), None)[1])
    return t


# instrument the model's pieces with forward marks and gradient taps
orig_rgb_local = model._rgb_local


def rgb_local(inputs, b):
    mark("rgb fwd start (rgb stream)")
    out = orig_rgb_local(inputs, b)
    mark("rgb fwd done (rgb stream)")
    return tap(out, "d rgb_local ready: rgb bwd can start")


model._rgb_local = rgb_local
for name in ("pts_cam_extractor", "implicit_transform", "main_estimator", "cam_enhancer"):
    mod = getattr(model, name)
    fwd = mod.forward

    def wrapped(*a, _f=fwd, _n=name, **k):
        mark(f"{_n} fwd start")
        out = _f(*a, **k)
        mark(f"{_n} fwd done")
        first = out[0] if isinstance(out, (tuple, list)) else out
        if torch.is_tensor(first):
            tap(first, f"{_n} bwd start")
        for t in a:
            if torch.is_tensor(t) and t.requires_grad and t.dim() == 3 and t.shape[1] in (128, 256):
                tap(t, f"{_n} bwd reached its feature input")
                break
        return out
    mod.forward = wrapped
wext = model.world_enhancer.extractor
wf = wext.forward


def wext_fwd(*a, **k):
    mark("world (GT) extractor fwd start")
    out = wf(*a, **k)
    mark("world (GT) extractor fwd done")
    return tap(out, "world (GT) extractor bwd start")


wext.forward = wext_fwd
first_conv = model.rgb_cam_extractor.model.feats.conv1


def step():
    mark("step start")
    opt.zero_grad(set_to_none=True)
    ep = model(batch)
    ep.update({k: batch[k] for k in ("rotation_label", "translation_label", "size_label", "qo")})
    loss = crit(ep)
    mark("loss fwd done")
    loss.backward()
    mark("backward returned (main stream)")
    opt.step()
    mark("optimizer done")


first_conv.weight.register_hook(lambda g: (mark("rgb bwd: first conv's weight gradient (rgb stream)"), None)[1])
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(4):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
eager = "--eager" in sys.argv
_native.MARKERS = {"buf": buf, "names": []}
if eager:
    step()
    names = list(_native.MARKERS["names"])
    _native.MARKERS = None
else:
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
print(f"# full-model step, {'eager' if eager else 'HIP-graph replay'}: in-stream markers (each marker is itself a ~3 us launch)")
print(f"{'t (ms)':>9}  marker")
for ti, n in rows:
    print(f"{(ti - t0) / 1e5:9.3f}  {n}")
