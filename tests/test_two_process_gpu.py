"""Two processes exchanging gradients of REAL training steps on the GPU (reference: nn.DataParallel, train.py:98-99).

The GPU box has one device, so both ranks use cuda:0 and the exchange runs over gloo (RCCL refuses two ranks on one
device); everything else is the N > 1 product path: one process per rank, identical replicas, local BatchNorm statistics,
the HIP kernels writing the flat gradient in place, ``OverlappedFlatReducer`` issuing its buckets from autograd hooks
(or after a graph replay), 1/N folded into the Adam launch.  No scaling number comes out of this and none is claimed."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import os, sys, torch
sys.path.insert(0, {root!r})
import torch.distributed as dist
import bench
from istnet_amd.optim import FlatAdam, layout_hints
from istnet_amd.parallel import OverlappedFlatReducer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
dist.init_process_group("gloo", rank=rank, world_size=world)
model = bench.make_model(dev, seed=0)                       # identical replicas
opt = FlatAdam(model.parameters(), lr=1e-3, adjacent=layout_hints(model))
red = OverlappedFlatReducer(opt, world, bucket_bytes=1 << 20)
issued = []
for it in range({steps}):
    pts = bench.shell_cloud(4, 1024, seed=10 * it + rank, device=dev)      # a different batch per rank and step
    opt.zero_grad(set_to_none=True)
    red.issued_in_backward = 0
    model(pts).square().mean().backward()
    issued.append(red.issued_in_backward)
    opt.step(red.finish(), grad_scale=1.0 / world)
torch.cuda.synchronize()
flat = opt.flat.detach().clone()
both = [torch.empty_like(flat) for _ in range(world)]
dist.all_gather(both, flat)
assert torch.equal(both[0], both[1]), "replicas diverged"
assert all(n == len(red.buckets) for n in issued), (issued, len(red.buckets))      # every bucket left DURING backward
if rank == 0:
    order = torch.tensor(opt.layout)
    torch.save({{"flat": flat.cpu(), "layout": order, "buckets": len(red.buckets)}}, {out!r})
dist.barrier()
dist.destroy_process_group()
print("OK", rank, issued)
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env(rank=None, world=2, port=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if rank is not None:
        env.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
    return env


def test_two_ranks_on_one_gpu_train_identically_and_match_hand_averaged_gradients(tmp_path):
    steps, port, out = 4, _free_port(), str(tmp_path / "rank0.pt")
    code = _WORKER.format(root=ROOT, steps=steps, out=out)
    procs = [subprocess.Popen([sys.executable, "-c", code], cwd=ROOT, env=_env(r, 2, port), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    results = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, results):
        assert p.returncode == 0 and "OK" in so, (so[-500:], se[-3000:])
    got = torch.load(out)

    # the same two steps in ONE process: each rank's batch differentiated at the current weights, the two flat gradients
    # summed (what the all-reduce does) and 1/2 applied inside the Adam launch -- the identical arithmetic, so bit-equal
    import bench
    from istnet_amd.optim import FlatAdam, layout_hints
    dev = torch.device("cuda:0")
    model = bench.make_model(dev, seed=0)
    opt = FlatAdam(model.parameters(), lr=1e-3, adjacent=layout_hints(model))
    assert torch.equal(torch.tensor(opt.layout), got["layout"])
    for it in range(steps):
        total = None
        for rank in range(2):
            pts = bench.shell_cloud(4, 1024, seed=10 * it + rank, device=dev)
            opt.zero_grad(set_to_none=True)
            model(pts).square().mean().backward()
            g = opt.pack_grads().clone()
            total = g if total is None else total + g
        opt.step(total, grad_scale=0.5)
    torch.cuda.synchronize()
    assert got["buckets"] >= 4
    assert torch.equal(opt.flat.cpu(), got["flat"]), float((opt.flat.cpu() - got["flat"]).abs().max())


@pytest.mark.parametrize("extra", [[], ["--eager", "--no-prefetch"]], ids=["graph_replay", "eager_hooks"])
def test_bench_two_ranks_same_device_reports_the_gradient_exchange(extra):
    """``bench.py --gpus 2`` on a 1-GPU box: both ranks on cuda:0, gloo exchange, the captured step (buckets after the
    replay) and the eager step (buckets from the hooks).  Checks the launch form, the exchange bookkeeping and that the
    JSON line is the driver's contract; the value is NOT a scaling measurement (two ranks share one GPU)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-device", "--backend", "gloo", "--steps", "3",
           "--warmup", "2", "--windows", "1", "--no-roofline", "--no-cpu-baseline", "--no-eager-leg"] + extra
    proc = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, (proc.stdout[-500:], proc.stderr[-3000:])
    line = [l for l in proc.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["global_batch"] == 64
    ex = d["config"]["gradient_exchange"]
    assert ex["bytes_per_step"] > 5_000_000 and ex["buckets"] >= 1 and sum(ex["bucket_bytes"]) == ex["bytes_per_step"]
    assert d["config"]["launch"] == ("eager" if extra else "hipgraph")
