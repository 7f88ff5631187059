import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda:0")
for bm in (False, True):
    torch.backends.cudnn.benchmark = bm
    net = bench.make_istnet(dev, 0).eval()
    batch = bench.istnet_batch(64, 2048, 0, dev)
    def step():
        with torch.no_grad():
            return net(batch)["pred_rotation"].cpu()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    print(f"cudnn.benchmark={bm}: {(time.perf_counter()-t0)/10*1e3:.2f} ms/batch", flush=True)
    # RGB branch alone
    ext = net.rgb_cam_extractor
    def rgb():
        with torch.no_grad():
            return ext(batch["rgb"], batch["choose"])
    for _ in range(3): rgb()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): rgb()
    torch.cuda.synchronize()
    print(f"   RGB branch alone: {(time.perf_counter()-t0)/10*1e3:.2f} ms", flush=True)
