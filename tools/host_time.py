import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
model = bench.make_model(dev); pts = bench.shell_cloud(32, 1024, 0, dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
step = bench.make_step(model, pts, opt, 1)  # eager encoder step
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/20:.2f} ms/step, total {1e3*(t2-t0)/20:.2f} ms/step")
