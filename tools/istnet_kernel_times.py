"""Device time of one eager full-model (config 3) training step by kernel name, with the step's wall time and the busy
time per stream -- the table the kernel-trace of rocprofv3 would give (it does not survive this process's exit)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from istnet_amd.optim import FlatAdam, layout_hints

from istnet_amd import tuned_gemm

tuned_gemm.enable()          # as bench.py --workload istnet / infer does
if "--infer" not in sys.argv:
    from istnet_amd.ist_net import point_branch_side_streams
    point_branch_side_streams(False)
dev = torch.device("cuda:0")
model = bench.make_istnet(dev)
if "--infer" in sys.argv:          # config 5: eval mode, B=64 N=2048, post-processing and the copy to the host included
    from istnet_amd import postprocess
    model.eval()
    batch = bench.istnet_batch(64, 2048, seed=0, device=dev)

    def step():
        with torch.no_grad():
            ep = model(batch)
            rts, scales = postprocess.assemble_pred_RTs(ep["pred_rotation"], ep["pred_translation"], ep["pred_size"])
            return rts.cpu(), scales.cpu()
else:
    fwd = [bench.make_istnet_fwd_bwd(model, bench.istnet_batch(32, 1024, 0, dev))]
    opt = FlatAdam(model.parameters(), lr=1e-4, adjacent=layout_hints(model))
    step = bench.make_graphed_step(fwd, opt, 1) if "--graph" in sys.argv else bench.make_eager_step(fwd, opt, 1)
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 5 * 1e3
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
agg, tl = {}, []
for ev in prof.events():
    if ev.device_type.name != "CUDA":
        continue
    t = ev.device_time_total if hasattr(ev, "device_time_total") else ev.cuda_time_total
    c, s = agg.get(ev.name, (0, 0.0))
    agg[ev.name] = (c + 1, s + t)
    tl.append((ev.time_range.start, ev.time_range.end, getattr(ev, "device_index", 0), getattr(ev, "device_resource_id", -1), ev.name))


def cat(n):
    if "anonymous namespace" in n and "at::native" not in n:
        return "library (libistnet_pn2)"
    if n.startswith("igemm_") or "MIOpen" in n or "SubTensorOp" in n or "ck::" in n or "_ZN2ck" in n:
        return "MIOpen"
    if n.startswith("Cijk_") or "rocblas" in n:
        return "hipBLASLt / rocBLAS"
    if n.startswith("Mem"):
        return "memcpy / memset"
    return "ATen"


total = sum(s for _, s in agg.values())
print(f"# full model eager {'inference batch' if '--infer' in sys.argv else 'training step'}: {wall:.2f} ms wall; sum of device time {total / 1e3:.2f} ms in {sum(c for c, _ in agg.values())} activities")
cats = {}
for n, (c, s) in agg.items():
    k = cat(n)
    c0, s0 = cats.get(k, (0, 0.0))
    cats[k] = (c0 + c, s0 + s)
for k, (c, s) in sorted(cats.items(), key=lambda kv: -kv[1][1]):
    print(f"# {k:28s} {c:5d} launches {s / 1e3:8.2f} ms")
print("   pct  calls   total_us    avg_us  kernel")
for n, (c, s) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:90]:
    print(f"{100 * s / total:6.2f} {c:6d} {s:10.1f} {s / c:9.1f}  {n[:200]}")


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")
    return n.split("(")[0][:44]


if "--timeline" in sys.argv and tl:
    tl.sort()
    t_first = tl[0][0]
    streams = sorted({e[3] for e in tl})
    print(f"# timeline: {len(streams)} device streams {streams}; one row per 0.5 ms: per stream busy fraction and the kernel with the most time")
    span = max(e[1] for e in tl) - t_first
    nb = int(span / 500) + 1
    for bi in range(nb):
        lo, hi = t_first + bi * 500, t_first + (bi + 1) * 500
        cells = []
        for sid in streams:
            busy, by = 0.0, {}
            for a, b, _, s_, n in tl:
                if s_ != sid or b <= lo or a >= hi:
                    continue
                d = min(b, hi) - max(a, lo)
                busy += d
                by[short(n)] = by.get(short(n), 0.0) + d
            top = max(by.items(), key=lambda kv: kv[1])[0] if by else ""
            cells.append(f"{busy / 500:4.2f} {top:44s}")
        print(f"{bi * 0.5:5.1f} | " + " | ".join(cells))
