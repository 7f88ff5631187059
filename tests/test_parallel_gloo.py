"""World-size-2 data-parallel path on CPU (gloo): the gradient exchange used by bench.py --gpus N."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make(seed=0):
    from istnet_amd.pointnet2.pointnet2_modules import PointnetFPModule, PointnetSAModuleMSG
    torch.manual_seed(seed)
    sa = PointnetSAModuleMSG(npoint=32, radii=[0.2, 0.4], nsamples=[8, 16], mlps=[[4, 8, 16], [4, 8, 16]])
    fp = PointnetFPModule(mlp=[32 + 4, 16])
    return torch.nn.ModuleList([sa, fp])


def _local_grads(model, rank):
    g = torch.Generator().manual_seed(100 + rank)
    xyz = torch.rand(2, 128, 3, generator=g)
    feat = torch.randn(2, 4, 128, generator=g)
    model.zero_grad()
    nx, nf = model[0](xyz, feat)
    model[1](xyz, nx, feat, nf).square().mean().backward()
    return [p.grad.clone() for p in model.parameters()]


def _worker(rank, world, port, bucket_bytes, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import istnet_amd  # noqa: F401
    from istnet_amd.parallel import GradAllReducer, broadcast_parameters
    from istnet_amd.pointnet2 import pointnet2_utils
    from oracle import pn2_oracle
    pointnet2_utils._ext = pn2_oracle     # CPU test harness only
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _make(seed=rank)              # deliberately different init per rank ...
    broadcast_parameters(model, src=0)    # ... made identical by the broadcast
    ref = _make(seed=0)
    same = all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), ref.state_dict().values()))
    _local_grads(model, rank)
    GradAllReducer(model, world, bucket_bytes=bucket_bytes).sync()
    want = [sum(gs) / world for gs in zip(*[_local_grads(_make(0), r) for r in range(world)])]
    ok = all(torch.allclose(p.grad, w, rtol=1e-5, atol=1e-7) for p, w in zip(model.parameters(), want))
    out[rank] = bool(same and ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes", [64 << 20, 2048])
def test_grad_allreduce_world2(bucket_bytes):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), bucket_bytes, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def _flat_worker(rank, world, port, out):
    """bench.py's N>1 optimizer path: FlatAdam.pack_grads -> GradAllReducer.sum_ -> FlatAdam.step(flat, 1/world)
    (first step), and the average_ spelling of the same exchange (second step)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import istnet_amd  # noqa: F401
    from istnet_amd.optim import FlatAdam
    from istnet_amd.parallel import GradAllReducer
    from istnet_amd.pointnet2 import pointnet2_utils
    from oracle import pn2_oracle
    pointnet2_utils._ext = pn2_oracle     # CPU test harness only
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _make(seed=0)
    opt = FlatAdam(model.parameters(), lr=1e-2)
    reducer = GradAllReducer(model, world)
    for it in range(2):
        opt.zero_grad(set_to_none=True)
        _local_grads_keep(model, rank)
        if it == 0:
            opt.step(reducer.sum_(opt.pack_grads()), grad_scale=1.0 / world)
        else:
            opt.step(reducer.average_(opt.pack_grads()))
    # single-process replay with the averaged gradients of both ranks
    ref = _make(seed=0)
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    for _ in range(2):
        per_rank = []
        for r in range(world):
            ref.zero_grad(set_to_none=True)
            _local_grads_keep(ref, r)
            per_rank.append([p.grad.clone() for p in ref.parameters()])
        for p, gs in zip(ref.parameters(), zip(*per_rank)):
            p.grad = sum(gs) / world
        ropt.step()
    ok = all(torch.allclose(p, q, rtol=1e-4, atol=1e-6) for p, q in zip(model.parameters(), ref.parameters()))
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    out[rank] = bool(ok and torch.equal(gathered[0], gathered[1]))   # replicas stay bit-identical
    dist.barrier()
    dist.destroy_process_group()


def _local_grads_keep(model, rank):
    g = torch.Generator().manual_seed(100 + rank)
    xyz = torch.rand(2, 128, 3, generator=g)
    feat = torch.randn(2, 4, 128, generator=g)
    nx, nf = model[0](xyz, feat)
    model[1](xyz, nx, feat, nf).square().mean().backward()


def test_flat_adam_allreduce_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_flat_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def _overlap_worker(rank, world, port, out):
    """parallel.OverlappedFlatReducer: buckets of FlatAdam.flat_grad all-reduced from post-accumulate-grad hooks while
    backward runs; two steps, the second with a parameter that receives no gradient."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import istnet_amd  # noqa: F401
    from istnet_amd.optim import FlatAdam
    from istnet_amd.parallel import OverlappedFlatReducer
    from istnet_amd.pointnet2 import pointnet2_utils
    from oracle import pn2_oracle
    pointnet2_utils._ext = pn2_oracle     # CPU test harness only
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _make(seed=0)
    extra = torch.nn.Parameter(torch.ones(7))          # a parameter no loss depends on
    opt = FlatAdam(list(model.parameters()) + [extra], lr=1e-2)
    red = OverlappedFlatReducer(opt, world, bucket_bytes=1024)      # several buckets
    assert len(red.buckets) >= 3 and red.buckets[0][1] == opt.flat_grad.numel() and red.buckets[-1][0] == 0
    assert sum(len(b[2]) for b in red.buckets) == len(opt.params)
    issued = []
    for it in range(2):
        opt.zero_grad(set_to_none=True)
        red.issued_in_backward = 0
        _local_grads_keep(model, rank)
        issued.append(red.issued_in_backward)
        opt.step(red.finish(), grad_scale=1.0 / world)
    ref = _make(seed=0)
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    for _ in range(2):
        per_rank = []
        for r in range(world):
            ref.zero_grad(set_to_none=True)
            _local_grads_keep(ref, r)
            per_rank.append([p.grad.clone() for p in ref.parameters()])
        for p, gs in zip(ref.parameters(), zip(*per_rank)):
            p.grad = sum(gs) / world
        ropt.step()
    ok = all(torch.allclose(p, q, rtol=1e-4, atol=1e-6) for p, q in zip(model.parameters(), ref.parameters()))
    ok = ok and torch.equal(extra.detach(), torch.ones(7))          # zero gradient: Adam leaves it alone
    # every bucket but the one holding the unused parameter (the last of the buffer, i.e. bucket 0) went out in backward
    ok = ok and all(n == len(red.buckets) - 1 for n in issued)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    out[rank] = bool(ok and torch.equal(gathered[0], gathered[1]))
    dist.barrier()
    dist.destroy_process_group()


def _reducer_validity_worker(rank, world, port, out):
    """Gradient validity is tracked by the reducer, not through ``p.grad is None``:
    (a) a captured step's gradients are read from the tensors recorded in its token even after zero_grad / with stale
        ``.grad`` attributes; (b) a second backward after a bucket left raises; (c) overlap=False accumulates."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import istnet_amd  # noqa: F401
    from istnet_amd.optim import FlatAdam
    from istnet_amd.parallel import OverlappedFlatReducer
    from istnet_amd.pointnet2 import pointnet2_utils
    from oracle import pn2_oracle
    pointnet2_utils._ext = pn2_oracle     # CPU test harness only
    dist.init_process_group("gloo", rank=rank, world_size=world)
    want = [sum(gs) for gs in zip(*[_local_grads(_make(0), r) for r in range(world)])]
    want_flat = torch.cat([w.reshape(-1) for w in want])
    ok = True
    # (a) token path: the "graph" wrote its gradients into private static tensors; .grad is None (zero_grad before the
    # replay) for half of the parameters and points at another graph's stale tensor for the rest
    model = _make(seed=0)
    opt = FlatAdam(model.parameters(), lr=1e-2)
    red = OverlappedFlatReducer(opt, world, bucket_bytes=1024)
    static = _local_grads(model, rank)                      # fires the hooks eagerly: buckets leave ...
    red.finish()                                            # ... and the step is closed
    token = {i: g for i, g in enumerate(static)}
    for i, p in enumerate(opt.params):
        p.grad = None if i % 2 else torch.full_like(p, 123.0)
    flat = red.finish(captured=token)
    ok = ok and torch.allclose(flat, want_flat, rtol=1e-5, atol=1e-7)
    # (b) accumulation with overlapped launches: the second backward must raise, not be silently dropped
    opt.zero_grad(set_to_none=True)
    _local_grads_keep(model, rank)
    raised = False
    try:
        _local_grads_keep(model, rank)
    except RuntimeError as exc:
        raised = "second backward" in str(exc)
    ok = ok and raised
    red.finish()
    # (c) overlap=False: every launch deferred to finish(), two backward passes accumulate
    model2 = _make(seed=0)
    opt2 = FlatAdam(model2.parameters(), lr=1e-2)
    red2 = OverlappedFlatReducer(opt2, world, bucket_bytes=1024, overlap=False)
    opt2.zero_grad(set_to_none=True)
    _local_grads_keep(model2, rank)
    _local_grads_keep(model2, rank)
    flat2 = red2.finish()
    ok = ok and torch.allclose(flat2, 2.0 * want_flat, rtol=1e-5, atol=1e-7)
    # (d) capture_collectives: a backward pass that is being captured issues its buckets from the hooks (they become graph
    # nodes on the GPU); finish_captured() issues the rest (the unused parameter's bucket), joins, returns the summed buffer.
    # No HIP graph on CPU: the hook's capture test is stood in for, the collectives run as they are issued.
    model3 = _make(seed=0)
    extra3 = torch.nn.Parameter(torch.ones(5))                # never used: no hook fires, its slot must come back zero
    opt3 = FlatAdam(list(model3.parameters()) + [extra3], lr=1e-2)
    red3 = OverlappedFlatReducer(opt3, world, bucket_bytes=1024, capture_collectives=True)
    red3._in_capture = lambda param: True
    red3.begin_capture()
    opt3.zero_grad(set_to_none=True)
    _local_grads_keep(model3, rank)
    launched_in_backward = sum(red3._cap_launched)
    flat3 = red3.finish_captured().clone()
    token3 = red3.end_capture()
    ok = ok and launched_in_backward == len(red3.buckets) - 1 and len(token3) == len(opt3.params) - 1
    ok = ok and torch.allclose(flat3[:want_flat.numel()], want_flat, rtol=1e-5, atol=1e-7)
    ok = ok and bool((flat3[want_flat.numel():] == 0).all())
    # (e) ADVICE round 4: an EAGER backward on the same reducer after a capture.  The capture's _launch() calls marked every
    # bucket "launched"; left that way, finish() would skip the bucket whose hooks do not all fire (the unused parameter's):
    # its slice neither zeroed nor all-reduced.  Poison the slot to see it.
    red3._in_capture = lambda param: False
    opt3.zero_grad(set_to_none=True)
    opt3.flat_grad.fill_(777.0)
    _local_grads_keep(model3, rank)
    flat3e = red3.finish()
    ok = ok and torch.allclose(flat3e[:want_flat.numel()], want_flat, rtol=1e-5, atol=1e-7)
    ok = ok and bool((flat3e[want_flat.numel():] == 0).all())
    raised = False
    try:
        OverlappedFlatReducer(FlatAdam(_make(0).parameters(), lr=1e-2), world).finish_captured()
    except RuntimeError as exc:
        raised = "capture_collectives" in str(exc)
    ok = ok and raised
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_reducer_tracks_gradient_validity_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_reducer_validity_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_overlapped_bucket_allreduce_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_overlap_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def _run_bench(cmd):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    proc = subprocess.run([sys.executable] + cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one JSON line, got {lines!r}"
    return json.loads(lines[0])


def test_bench_self_launches_n_ranks():
    """``python bench.py --gpus 2 ...`` (the driver's N=1 command form with N=2, no launcher, WORLD_SIZE unset)
    spawns its own two ranks and rank 0 prints the one JSON line.  --cpu-dry-run keeps the control flow of the GPU
    run (rendezvous, flat-gradient all-reduce, barriers, max-over-ranks timing) on host cores."""
    res = _run_bench(["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-dry-run", "--capture-allreduce"])
    # (--capture-allreduce: the flag of the in-graph exchange; a CPU dry run has no graph, the eager form must still run)
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["warmup"] == 1
    assert res["config"]["parallelism"] == "dp2" and res["scaling"] == "weak"
    assert res["config"]["global_batch"] == 2 * res["config"]["batch_per_gpu"]
    assert res["value"] > 0 and "NOT a measurement" in res["data"]


def test_bench_under_torch_distributed_run():
    """The driver's N>1 command form: ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W``."""
    res = _run_bench(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                      "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "1",
                      "--warmup", "1", "--cpu-dry-run"])
    assert res["n_gpus"] == 2 and res["config"]["parallelism"] == "dp2"


def test_bench_istnet_workload_two_ranks_overlapped_exchange():
    """``--workload istnet`` with N = 2 (toy size on host cores): the full model's gradients leave in several buckets FROM
    THE AUTOGRAD HOOKS while backward runs (``--overlap-allreduce``: the eager step; the default on GPUs is the captured
    step with the buckets after the replay), and the JSON says how many bytes and buckets a step exchanges."""
    res = _run_bench(["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1", "--cpu-dry-run", "--workload", "istnet",
                      "--overlap-allreduce"])
    assert res["n_gpus"] == 2 and res["config"]["parallelism"] == "dp2" and res["config"]["launch"] == "eager"
    ex = res["config"]["gradient_exchange"]
    assert ex["buckets"] >= 4 and sum(ex["bucket_bytes"]) == ex["bytes_per_step"]
    assert 100e6 < ex["bytes_per_step"] < 115e6          # 26.8 M parameters in fp32 (107.3 MB)
    assert "during backward" in ex["issued"]
    assert res["value"] > 0 and "NOT a measurement" in res["data"]
