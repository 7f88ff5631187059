"""Where a launch of pw_fwd_sk_kernel spends its time: cycles per phase (thread 0 of every workgroup, clock64) summed over the
launches of a few encoder training steps.  Needs a library built with -DISTNET_PHASE_TIMING (tools/gpu_phase.sh builds
nothing on the GPU box: the instrumented library is built beforehand as ab_base/phase.so and copied over the product's)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from istnet_amd import _native
from istnet_amd.optim import FlatAdam, layout_hints
from istnet_amd.modules import GeometrySlot

dev = torch.device("cuda:0")
model = bench.make_model(dev)
batches = [bench.shell_cloud(32, 1024, s, dev) for s in (0, 1000)]
slots = [model.prefetch_geometry(bt, GeometrySlot()) for bt in batches]
fwd = [bench.make_pipelined_fwd_bwd(model, batches, slots, i) for i in (0, 1)]
opt = FlatAdam(model.parameters(), lr=1e-4, adjacent=layout_hints(model))
step = bench.make_eager_step(fwd, opt, 1)
for _ in range(4):
    step()
torch.cuda.synchronize()
lib = ctypes.CDLL(_native.LIB_PATH)
out = (ctypes.c_ulonglong * 18)()
assert lib.istnet_debug_phase_read(out, 1) == 0
STEPS = 10
for _ in range(STEPS):
    step()
torch.cuda.synchronize()
assert lib.istnet_debug_phase_read(out, 0) == 0
names = ["BatchNorm constants -> LDS (+ barrier)", "first operand group in registers", "K loop (wave 0)",
         "barrier: the other waves finish", "split-K exchange through LDS (+ barrier)",
         "epilogue: four rows formed and stored (the timestamp drains this wave's older stores)", "remaining stores acknowledged"]
for kind, label in ((0, "plain launches (source products, layer >= 1)"), (1, "launches with the three_interpolate epilogue (FP layer 0)")):
    wgs = out[16 + kind]
    tot = sum(out[8 * kind + i] for i in range(7))
    print(f"# pw_fwd_sk_kernel, {label}: {wgs / STEPS:.0f} workgroups per step; cycles per workgroup (thread 0, clock64), {STEPS} eager steps")
    for i, n in enumerate(names):
        print(f"{out[8 * kind + i] / max(wgs, 1):9.0f} cycles  {100 * out[8 * kind + i] / max(tot, 1):5.1f} %  {n}")
    print(f"{tot / max(wgs, 1):9.0f} cycles per workgroup")
