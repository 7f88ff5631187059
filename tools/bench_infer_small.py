"""Inference at the batch sizes of the reference's test loop (test.py / utils/solver.py:199-262 feed ONE IMAGE per step: B = its
instances): the model called exactly as there -- eval mode, torch.no_grad(), ``model(inputs)``, results copied to the host --
launch by launch (ISTNET_AUTO_GRAPH=0) and with the transparent per-batch-size graphs (graphed.InferenceGraph, the default).
    python tools/bench_infer_small.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from istnet_amd import graphed, postprocess

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda:0")
net = bench.make_istnet(dev, seed=0).eval()


def timeit(fn, n=40):
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"# IST-Net inference, N={N}, 192x192 crops; ms per image (forward + assemble_pred_RTs + copy to the host), 1 x MI355X")
for B in (1, 2, 3, 4, 6, 8, 12, 16, 32, 64):
    batch = bench.istnet_batch(B, N, seed=B, device=dev)

    def image():
        with torch.no_grad():
            ep = net(batch)
            rts, sc = postprocess.assemble_pred_RTs(ep["pred_rotation"], ep["pred_translation"], ep["pred_size"])
            return rts.cpu(), sc.cpu()

    graphed.ENABLED = False
    ms_plain = timeit(image)
    ref = image()
    graphed.ENABLED = True
    ms_graph = timeit(image)
    got = image()
    diff = max(float((a - b).abs().max()) for a, b in zip(got, ref))
    print(f"B={B:3d}: launch by launch {ms_plain:7.3f}   graphs {ms_graph:7.3f}   x{ms_plain / ms_graph:.2f}   "
          f"{B / ms_graph * 1e3:7.0f} instances/s   max |pose diff| {diff:.1e}", flush=True)
print("# graphs:", {k: v for k, v in graphed.STATS.items() if k.startswith("infer")},
      f" memory held {torch.cuda.memory_reserved() / 2**30:.1f} GiB")
