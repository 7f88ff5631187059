"""Per-stage device time of the input preparation (bench.py --workload pipeline), B=32 training and B=64 inference frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from istnet_amd import preprocess

dev = torch.device("cuda:0")


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) / n * 1e3, (t1 - t0) / n * 1e3, r


for b, n in ((32, 1024), (64, 2048)):
    fr = bench.synthetic_frames(b, n, 0, dev)
    print(f"# B={b} N={n}: ms per batch (device-complete, host-issue)")
    ms, host, depth = t(lambda: preprocess.fill_missing(fr["depth"], 1000.0, 1)); print(f"fill_missing            {ms:7.3f} {host:7.3f}")
    ms, host, win = t(lambda: preprocess.get_bbox(fr["boxes"])); print(f"get_bbox                {ms:7.3f} {host:7.3f}")
    ms, host, rgb = t(lambda: preprocess.crop_resize_normalize(fr["image"], win, 192)); print(f"crop_resize_normalize   {ms:7.3f} {host:7.3f}")
    ms, host, _ = t(lambda: rgb.contiguous(memory_format=torch.channels_last)); print(f"to channels_last        {ms:7.3f} {host:7.3f}")
    ms, host, (pts, ch) = t(lambda: preprocess.backproject_choose(depth, win, fr["choose"])); print(f"backproject_choose      {ms:7.3f} {host:7.3f}")
    g = torch.Generator().manual_seed(0)
    ms, host, _ = t(lambda: preprocess.jitter_points(pts, generator=g)); print(f"jitter_points           {ms:7.3f} {host:7.3f}")
    rot = torch.linalg.qr(torch.randn(b, 3, 3))[0].to(dev)
    tr, sc, sz = torch.randn(b, 3, device=dev), torch.rand(b, device=dev), torch.rand(b, 3, device=dev)
    sym = torch.zeros(b, dtype=torch.bool, device=dev)
    ms, host, (r2, size, qo, _) = t(lambda: preprocess.instance_labels(pts, tr, rot, sc, sz, sym)); print(f"instance_labels         {ms:7.3f} {host:7.3f}")
    ms, host, (bb, rt_t, rt_r) = t(lambda: preprocess.generate_aug_parameters(b, device=dev, generator=g)); print(f"generate_aug_parameters {ms:7.3f} {host:7.3f}")
    model = torch.rand(b, 1024, 3, device=dev) - 0.5
    cat = torch.randint(0, 6, (b,), device=dev)
    symi = torch.zeros(b, 4, dtype=torch.long, device=dev)
    ms, host, _ = t(lambda: preprocess.data_augment(preprocess.AUG_PROBS_DEFAULT, pts, r2.float(), tr, size, symi, bb, rt_t, rt_r, model,
                                                    qo.float(), cat, generator=g)); print(f"data_augment            {ms:7.3f} {host:7.3f}")
