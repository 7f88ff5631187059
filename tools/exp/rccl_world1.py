import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.ones(1 << 20, device=dev)
dist.all_reduce(x); dist.barrier(); torch.cuda.synchronize()
print("rccl world=1 all_reduce ok", float(x.sum()))
# all-reduce inside a captured graph
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): dist.all_reduce(x)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
try:
    with torch.cuda.graph(g):
        y = x * 2
        dist.all_reduce(y)
    g.replay(); torch.cuda.synchronize()
    print("rccl all_reduce captured in a HIP graph: ok", float(y[0]))
except Exception as e:
    print("capture with all_reduce failed:", type(e).__name__, str(e)[:200])
dist.destroy_process_group()
