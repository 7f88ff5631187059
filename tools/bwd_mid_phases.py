"""Where pw_bwd_mid_kernel spends its time: cycles per phase of MFMA wave 0 and loader wave 4 of every workgroup (clock64),
summed over the launches of a few eager encoder steps, per template instance.  Needs a library built with
-DISTNET_PHASE_TIMING (built beforehand as ab_base/phase.so and copied over the product's library on the box)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from istnet_amd import _native
from istnet_amd.optim import FlatAdam, layout_hints

dev = torch.device("cuda:0")
model = bench.make_model(dev)
pts = bench.shell_cloud(32, 1024, 0, dev)
opt = FlatAdam(model.parameters(), lr=1e-4, adjacent=layout_hints(model))
step = bench.make_eager_step(bench.make_encoder_fwd_bwd(model, pts), opt, 1)
for _ in range(4):
    step()
torch.cuda.synchronize()
lib = ctypes.CDLL(_native.LIB_PATH)
out = (ctypes.c_ulonglong * 102)()
assert lib.istnet_debug_mid_phase_read(out, 1) == 0
STEPS = 10
for _ in range(STEPS):
    step()
torch.cuda.synchronize()
assert lib.istnet_debug_mid_phase_read(out, 0) == 0
kinds = ["<8,4,*,false> (cout 256, dgrad only)", "<4,4> (128 -> 128)", "<4,2> (64 -> 128)", "<2,2> (64 -> 64)", "<2,1> (32 -> 64)", "other"]
for k, name in enumerate(kinds):
    wgs = out[96 + k]
    if not wgs:
        continue
    c = [out[(k * 2 + 0) * 8 + i] / wgs for i in range(8)]
    l = [out[(k * 2 + 1) * 8 + i] / wgs for i in range(8)]
    print(f"# pw_bwd_mid_kernel{name}: {wgs / STEPS:.0f} workgroups per step; cycles per workgroup")
    print(f"  MFMA wave 0 : barrier wait {c[0]:9.0f}   compute (MFMAs + epilogue) {c[1]:9.0f}   -> {100 * c[0] / max(c[0] + c[1], 1):4.1f} % waiting")
    print(f"  loader wave : first loads {l[0]:8.0f}   store_chunk (wait loads + dY + LDS) {l[1]:9.0f}   issue next loads {l[2]:8.0f}   barrier wait {l[3]:9.0f}")
