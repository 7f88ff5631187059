"""Full-model inference of BASELINE config 5 (eval mode, B=64, N=2048, 192x192 crops): model + the post-processing of
test_func (solver.py:231-241) + the device->host copy of the result, with the RGB tail dense (reference order:
final layer on all H*W pixels, then the `choose` gather) and gather-first (final layer on the chosen pixels only).
Prints ms/batch, instances/s and the pose deltas between the two orders.     python tools/bench_infer_full.py [B] [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from istnet_amd import ist_net, postprocess

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
dev = torch.device("cuda:0")
net = bench.make_istnet(dev, seed=0)
# non-trivial BatchNorm running statistics, as after training
g = torch.Generator().manual_seed(5)
for m in net.modules():
    if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
        m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
        m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
net.eval()
batch = bench.istnet_batch(B, N, seed=0, device=dev)


def infer():
    with torch.no_grad():
        ep = net(batch)
        rts, scales = postprocess.assemble_pred_RTs(ep["pred_rotation"], ep["pred_translation"], ep["pred_size"])
        return rts.cpu(), scales.cpu(), ep


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


res = {}
for tag, flag in (("dense tail (reference order)", False), ("gather-first tail", True)):
    ist_net.USE_GATHER_FIRST = flag
    ms = timeit(infer)
    res[tag] = infer()
    print(f"{tag:32s} B={B} N={N}: {ms:8.2f} ms/batch  {B / ms * 1e3:8.0f} instances/s  "
          f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    torch.cuda.reset_peak_memory_stats()
a, b = res.values()
print("max |delta| between the two orders: pred_RTs %.3e  pred_scales %.3e  pred_qo %.3e" % (
    (a[0] - b[0]).abs().max(), (a[1] - b[1]).abs().max(), (a[2]["pred_qo"] - b[2]["pred_qo"]).abs().max()))
ist_net.USE_GATHER_FIRST = True
with torch.no_grad():
    ep = net(batch)
    rts, _ = postprocess.assemble_pred_RTs(ep["pred_rotation"], ep["pred_translation"], ep["pred_size"])
    gt = rts.double().clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        err = postprocess.pose_errors(rts, gt, batch["category_label"].reshape(-1) + 1, torch.ones(B, device=dev))
    torch.cuda.synchronize()
    cpu_args = (rts.cpu(), gt.cpu(), (batch["category_label"].reshape(-1) + 1).cpu(), torch.ones(B))
    t1 = time.perf_counter()
    for _ in range(20):
        postprocess.pose_errors(*cpu_args)
    print(f"pose_errors {B}x{B} table on the host: {(time.perf_counter() - t1) / 20 * 1e3:.3f} ms")
    print(f"pose_errors {B}x{B} table on the GPU: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms; "
          f"diagonal max (non-symmetric classes) {err.diagonal(dim1=0, dim2=1)[0].nan_to_num().max():.2e} deg")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    infer()
    torch.cuda.synchronize()
evs = sorted((e for e in prof.key_averages() if e.device_time_total > 0), key=lambda e: -e.device_time_total)
total = sum(e.device_time_total for e in evs)
print(f"kernels in one forward: {sum(e.count for e in evs)}, {total / 1e3:.2f} ms of kernel time; top:")
for e in evs[:12]:
    print(f"  {e.device_time_total:9.1f} us  x{e.count:4d}  {e.key[:100]}")
