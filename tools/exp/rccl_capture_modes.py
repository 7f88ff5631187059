"""Does a pending-work poll of the ProcessGroupNCCL watchdog thread abort the process while ANOTHER thread captures a
HIP graph?  world_size 1 on one GPU.   python tools/exp/rccl_capture_modes.py global|thread_local|relaxed"""
import os, sys, time, torch, torch.distributed as dist
mode = sys.argv[1]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.ones(1 << 20, device=dev)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(10): dist.all_reduce(x)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode=mode):
    y = x * 2
    time.sleep(3.0)          # several watchdog polls happen while the capture is open
    z = y + 1
g.replay(); dist.all_reduce(z); torch.cuda.synchronize()
print(f"capture mode {mode}: survived, z[0] = {float(z[0])}")
dist.destroy_process_group()
