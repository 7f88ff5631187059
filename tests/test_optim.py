"""FlatAdam against torch.optim.Adam on the same parameters and gradients (CPU; the GPU path is the same fused
kernel torch.optim.Adam(fused=True) launches, checked in test_optim_gpu)."""
import pytest
import torch

import istnet_amd  # noqa: F401
from istnet_amd.optim import FlatAdam


def _models():
    torch.manual_seed(0)
    a = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    b = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    b.load_state_dict(a.state_dict())
    return a, b


def _run(dev, steps=5, wd=0.0):
    a, b = _models()
    a, b = a.to(dev), b.to(dev)
    ref = torch.optim.Adam(a.parameters(), lr=1e-2, weight_decay=wd)
    opt = FlatAdam(b.parameters(), lr=1e-2, weight_decay=wd)
    g = torch.Generator().manual_seed(1)
    for _ in range(steps):
        x = torch.randn(4, 5, generator=g).to(dev)
        for m, o in ((a, ref), (b, opt)):
            o.zero_grad(set_to_none=True)
            m(x).square().mean().backward()
            o.step()
    for p, q in zip(a.parameters(), b.parameters()):
        torch.testing.assert_close(q, p, rtol=1e-5, atol=1e-7)


def test_flat_adam_matches_torch_adam_cpu():
    _run("cpu")
    _run("cpu", wd=0.01)


def test_flat_adam_parameters_are_views_and_packed_grads():
    _, b = _models()
    opt = FlatAdam(b.parameters(), lr=1e-2)
    assert all(p.data_ptr() == opt.flat.data_ptr() + 4 * o for p, o in zip(opt.params, opt.offsets))
    b(torch.ones(2, 5)).sum().backward()
    flat = opt.pack_grads()
    for v, p in zip(opt.grad_views(flat), opt.params):
        assert torch.equal(v, p.grad)
    before = opt.flat.clone()
    opt.step(flat)                      # pre-packed gradients
    assert not torch.equal(before, opt.flat)


def test_flat_adam_adjacent_layout_keeps_the_parameter_order_of_the_state():
    """``adjacent`` lays the named parameters out back to back (a view of the buffer is then their stacked matrix) while
    updates, packed gradients and the torch-Adam state_dict stay those of the parameter order."""
    from istnet_amd.optim import layout_hints
    from istnet_amd.pointnet2.pointnet2_modules import PointnetSAModuleMSG
    torch.manual_seed(5)
    a = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 3), torch.nn.Linear(3, 7))
    b = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 3), torch.nn.Linear(3, 7))
    b.load_state_dict(a.state_dict())
    ref = torch.optim.Adam(a.parameters(), lr=1e-2)
    opt = FlatAdam(b.parameters(), lr=1e-2, adjacent=[[b[0].bias, b[3].bias]])
    i0, i3 = 1, 5                                                   # indices of the two biases in parameter order
    assert opt.offsets[i3] == opt.offsets[i0] + 7 and opt.layout[:3] == [0, 1, 5]
    stacked = b[0].bias.detach().as_strided((2, 7), (7, 1))
    assert torch.equal(stacked[1], b[3].bias.detach())
    g = torch.Generator().manual_seed(0)
    for _ in range(4):
        x = torch.randn(4, 5, generator=g)
        for m, o in ((a, ref), (b, opt)):
            o.zero_grad(set_to_none=True)
            (m[2](torch.tanh(m[0](x))).square().mean() + m[3](m[2](torch.tanh(m[0](x)))).mean()).backward()
            o.step()
    for p, q in zip(a.parameters(), b.parameters()):
        torch.testing.assert_close(q, p, rtol=1e-5, atol=1e-7)
    sd, sd_ref = opt.state_dict(), ref.state_dict()
    for k in sd_ref["state"]:
        torch.testing.assert_close(sd["state"][k]["exp_avg"], sd_ref["state"][k]["exp_avg"], rtol=1e-5, atol=1e-8)
    flat = opt.pack_grads()
    for v, p in zip(opt.grad_views(flat), opt.params):
        assert torch.equal(v, p.grad)
    # the set-abstraction module offers its scales' layer-0 weights
    sa = PointnetSAModuleMSG(npoint=8, radii=[0.1, 0.2], nsamples=[4, 8], mlps=[[6, 8, 8], [6, 8, 16]])
    hints = layout_hints(sa)
    assert len(hints) == 1 and [tuple(w.shape) for w in hints[0]] == [(8, 9, 1, 1), (8, 9, 1, 1)]
    o2 = FlatAdam(sa.parameters(), adjacent=hints)
    w0a, w0b = hints[0]
    assert w0b.data_ptr() == w0a.data_ptr() + 4 * w0a.numel()


@pytest.mark.gpu
def test_flat_adam_matches_torch_adam_gpu():
    _run("cuda:0")
    _run("cuda:0", wd=0.01)


@pytest.mark.gpu
def test_fused_backward_writes_gradients_in_place():
    """With FlatAdam attached, the fused SA / FP backward kernels write every dW, dgamma, dbeta straight into the
    optimizer's flat gradient buffer (no pack), and the result equals the unattached run."""
    from istnet_amd.pointnet2.pointnet2_modules import PointnetFPModule, PointnetSAModuleMSG

    def make():
        torch.manual_seed(0)
        sa = PointnetSAModuleMSG(npoint=64, radii=[0.2, 0.4], nsamples=[16, 32], mlps=[[16, 32, 32], [16, 32, 64]])
        fp = PointnetFPModule(mlp=[96 + 16, 64, 32])
        return torch.nn.ModuleList([sa, fp]).cuda().train()

    g = torch.Generator().manual_seed(3)
    xyz = torch.rand(2, 256, 3, generator=g).cuda()
    feat = torch.randn(2, 16, 256, generator=g).cuda()

    def run(model):
        nx, nf = model[0](xyz, feat)
        model[1](xyz, nx, feat, nf).square().mean().backward()

    ref = make()
    run(ref)
    model = make()
    opt = FlatAdam(model.parameters(), lr=1e-3)
    run(model)
    torch.cuda.synchronize()
    assert opt.grads_in_place()
    assert opt.pack_grads().data_ptr() == opt.flat_grad.data_ptr()
    for p, q in zip(model.parameters(), ref.parameters()):
        torch.testing.assert_close(p.grad, q.grad, rtol=1e-5, atol=1e-7)
    run(model)                          # second backward without zero_grad: accumulates, still the same storage
    torch.cuda.synchronize()
    for p, q in zip(model.parameters(), ref.parameters()):
        torch.testing.assert_close(p.grad, 2 * q.grad, rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
def test_backward_after_native_step_raises():
    """forward -> step -> backward: the native update writes the flat buffer through raw pointers; FlatAdam.step bumps the
    parameters' version counters so that autograd refuses to differentiate against the already-updated weights (the fused
    nodes save views of live parameter memory) instead of doing it silently (pin the guard)."""
    from istnet_amd.pointnet2.pointnet2_modules import PointnetSAModuleMSG
    torch.manual_seed(0)
    sa = PointnetSAModuleMSG(npoint=64, radii=[0.2, 0.4], nsamples=[16, 32], mlps=[[16, 32, 32], [16, 32, 64]]).cuda().train()
    g = torch.Generator().manual_seed(3)
    xyz = torch.rand(2, 256, 3, generator=g).cuda()
    feat = torch.randn(2, 16, 256, generator=g).cuda()
    opt = FlatAdam(sa.parameters(), lr=1e-3)
    sa(xyz, feat)[1].square().mean().backward()
    versions = [p._version for p in sa.parameters()]
    stale = sa(xyz, feat)[1].square().mean()        # forward BEFORE the update ...
    opt.step()
    assert all(p._version == v + 1 for p, v in zip(sa.parameters(), versions))
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        stale.backward()                            # ... backward after it


@pytest.mark.gpu
def test_flat_adam_kernel_odd_sizes_and_grad_scale():
    """istnet_adam_step on lengths that are not multiples of 4 / 1024, with weight decay and a folded 1/world scale,
    against the plain-ops update in float64."""
    dev = torch.device("cuda:0")
    for n in (1, 3, 1021, 4096, 70001):
        g = torch.Generator().manual_seed(n)
        p0 = torch.randn(n, generator=g)
        model = [torch.nn.Parameter(p0.clone().to(dev))]
        opt = FlatAdam(model, lr=3e-3, betas=(0.8, 0.95), eps=1e-6, weight_decay=0.02)
        p, m, v = p0.double(), torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
        for t in range(1, 4):
            grad = torch.randn(n, generator=g)
            opt.step(grad.to(dev), grad_scale=0.5)
            gg = grad.double() * 0.5 + 0.02 * p
            m = 0.8 * m + 0.2 * gg
            v = 0.95 * v + 0.05 * gg * gg
            p = p - 3e-3 / (1 - 0.8 ** t) * m / (v.sqrt() / (1 - 0.95 ** t) ** 0.5 + 1e-6)
        torch.testing.assert_close(opt.flat.cpu().double(), p, rtol=2e-5, atol=1e-7)
        assert model[0].data_ptr() == opt.flat.data_ptr()


@pytest.mark.gpu
def test_lr_schedule_reaches_a_captured_step_without_recapture():
    """FlatAdam keeps the learning rate on the device: ``opt.lr = v`` between replays of ONE captured step changes
    the update exactly as it does for torch.optim.Adam with a per-iteration schedule (solver.py:88-89)."""
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ref_p = torch.nn.Parameter(torch.randn(1000, device=dev))
    my_p = torch.nn.Parameter(ref_p.detach().clone())
    ref = torch.optim.Adam([ref_p], lr=1e-2)
    opt = FlatAdam([my_p], lr=1e-2)
    grad = torch.randn(1000, device=dev)
    static_grad = torch.zeros_like(grad)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        opt.step(static_grad)                      # warm-up launch outside the capture (a zero-gradient step)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ref_p.grad = torch.zeros_like(grad)
    ref.step()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        opt.step(static_grad)
    for it, lr in enumerate([1e-2, 3e-3, 5e-2, 1e-4]):
        g = grad * (it + 1)
        static_grad.copy_(g)
        opt.lr = lr
        graph.replay()
        ref.param_groups[0]["lr"] = lr
        ref_p.grad = g.clone()
        ref.step()
    torch.testing.assert_close(opt.flat, ref_p.detach(), rtol=1e-5, atol=1e-7)
    assert opt.lr == 1e-4 and int(opt.step_count) == 5     # warm-up + four replays (the capture pass does not execute)


def test_flat_adam_works_with_torch_lr_schedulers():
    """FlatAdam is a torch.optim.Optimizer: CyclicLR (what utils/solver.py:46-47 builds) drives it like torch's Adam."""
    a, b = _models()
    ref = torch.optim.Adam(a.parameters(), lr=1e-3)
    opt = FlatAdam(b.parameters(), lr=1e-3)
    scheds = [torch.optim.lr_scheduler.CyclicLR(o, base_lr=1e-5, max_lr=1e-2, step_size_up=3, mode="triangular",
                                                cycle_momentum=False) for o in (ref, opt)]
    g = torch.Generator().manual_seed(2)
    for _ in range(8):
        x = torch.randn(4, 5, generator=g)
        for m, o, sch in ((a, ref, scheds[0]), (b, opt, scheds[1])):
            o.zero_grad(set_to_none=True)
            m(x).square().mean().backward()
            o.step()
            sch.step()
    assert abs(opt.param_groups[0]["lr"] - ref.param_groups[0]["lr"]) < 1e-12
    for p, q in zip(a.parameters(), b.parameters()):
        torch.testing.assert_close(q, p, rtol=1e-5, atol=1e-7)
    with pytest.raises(ValueError):
        FlatAdam([{"params": list(b.parameters())}])


def test_flat_adam_state_dict_round_trips_with_torch_adam():
    """FlatAdam.state_dict() has torch.optim.Adam's layout (the reference checkpoints hold optimizer.state_dict() of
    torch Adam, utils/solver.py): a torch Adam loads ours, ours loads torch's, training continues identically, and
    loading an optimizer state never touches the weights."""
    a, b = _models()
    ref = torch.optim.Adam(a.parameters(), lr=1e-2, weight_decay=0.01)
    opt = FlatAdam(b.parameters(), lr=1e-2, weight_decay=0.01)
    g = torch.Generator().manual_seed(7)

    def steps(pairs, n):
        for _ in range(n):
            x = torch.randn(4, 5, generator=g)
            for m, o in pairs:
                o.zero_grad(set_to_none=True)
                m(x).square().mean().backward()
                o.step()

    steps(((a, ref), (b, opt)), 3)
    sd_ref, sd_opt = ref.state_dict(), opt.state_dict()
    assert set(sd_opt) == {"state", "param_groups"} and set(sd_opt["state"]) == set(sd_ref["state"])
    for k in sd_ref["state"]:
        assert set(sd_opt["state"][k]) == set(sd_ref["state"][k])
        for name in ("exp_avg", "exp_avg_sq"):
            torch.testing.assert_close(sd_opt["state"][k][name], sd_ref["state"][k][name], rtol=1e-5, atol=1e-9)
        assert float(sd_opt["state"][k]["step"]) == float(sd_ref["state"][k]["step"]) == 3.0
    assert sd_opt["param_groups"][0]["params"] == sd_ref["param_groups"][0]["params"]
    # cross-load: fresh optimizers of each kind resume from the other kind's state
    a2, b2 = _models()
    a2.load_state_dict(a.state_dict()); b2.load_state_dict(b.state_dict())
    ref2 = torch.optim.Adam(a2.parameters(), lr=5e-1)
    opt2 = FlatAdam(b2.parameters(), lr=5e-1)
    weights_before = opt2.flat.clone()
    ref2.load_state_dict(sd_opt)
    opt2.load_state_dict(sd_ref)
    assert torch.equal(opt2.flat, weights_before)                     # an optimizer state carries no weights
    assert opt2.lr == 1e-2 and opt2.weight_decay == 0.01 and int(opt2.step_count) == 3
    steps(((a, ref), (b, opt), (a2, ref2), (b2, opt2)), 2)
    for p, q, r, s in zip(a.parameters(), b.parameters(), a2.parameters(), b2.parameters()):
        torch.testing.assert_close(q, p, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(r, p, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(s, p, rtol=1e-5, atol=1e-7)
    with pytest.raises(ValueError):
        FlatAdam(list(b.parameters())[:2], lr=1e-2).load_state_dict(sd_ref)


def test_flat_adam_state_dict_details_of_reference_checkpoints():
    """``step`` is saved as a CPU float tensor (torch.optim.Adam's non-capturable layout); keys a scheduler
    adds to the parameter group (``initial_lr``) survive a save / load round trip, so a scheduler resumed with
    last_epoch != -1 works; a checkpoint saved by Adam over ALL parameters of a model loads into an optimizer over the
    trainable ones (the frozen-world-enhancer stage, train.py:102-118) through ``all_params``."""
    a, b = _models()
    opt = FlatAdam(b.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.CyclicLR(opt, base_lr=1e-5, max_lr=1e-2, step_size_up=3, cycle_momentum=False)
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        b(torch.ones(4, 5)).square().mean().backward()
        opt.step(); sched.step()
    sd = opt.state_dict()
    assert all(v["step"].device.type == "cpu" and v["step"].dtype == torch.float32 for v in sd["state"].values())
    assert "initial_lr" in sd["param_groups"][0]
    _, b2 = _models()
    opt2 = FlatAdam(b2.parameters(), lr=1e-3)
    opt2.load_state_dict(sd)
    assert opt2.param_groups[0]["initial_lr"] == sd["param_groups"][0]["initial_lr"]
    torch.optim.lr_scheduler.CyclicLR(opt2, base_lr=1e-5, max_lr=1e-2, step_size_up=3, cycle_momentum=False, last_epoch=1)
    # a state saved over all parameters, loaded by an optimizer over the trainable subset
    m, _ = _models()
    full = torch.optim.Adam(m.parameters(), lr=1e-2)
    m(torch.ones(4, 5)).square().mean().backward()
    full.step()
    sd_full = full.state_dict()
    params = list(m.parameters())
    params[0].requires_grad_(False)
    sub = FlatAdam(m.parameters(), lr=1e-2)
    with pytest.raises(ValueError):
        sub.load_state_dict(sd_full)
    sub.all_params = params
    sub.load_state_dict(sd_full)
    torch.testing.assert_close(sub.exp_avg[:params[1].numel()].view_as(params[1]), sd_full["state"][1]["exp_avg"])


@pytest.mark.gpu
def test_shared_module_used_twice_in_one_graph_under_flat_adam():
    """A fused module applied twice in ONE autograd graph (siamese use): both backward nodes see ``param.grad is
    None``; only the first may take the parameter's slot of the flat gradient buffer, the engine adds the second
    producer's tensor.  Three optimizer steps against torch.optim.Adam on an identical copy."""
    from istnet_amd.modules import PointNet2MSG
    from istnet_amd.pointnet2.pointnet2_modules import PointnetFPModule, PointnetSAModuleMSG

    def make():
        torch.manual_seed(0)
        sa = PointnetSAModuleMSG(npoint=64, radii=[0.2, 0.4], nsamples=[16, 32], mlps=[[16, 32, 32], [16, 32, 64]])
        fp = PointnetFPModule(mlp=[96 + 16, 64, 32])
        return torch.nn.ModuleList([sa, fp]).cuda().train()

    g = torch.Generator().manual_seed(3)
    clouds = [(torch.rand(2, 256, 3, generator=g).cuda(), torch.randn(2, 16, 256, generator=g).cuda()) for _ in range(2)]

    def loss_of(model):
        total = 0.0
        for xyz, feat in clouds:                       # the same modules twice in one graph
            nx, nf = model[0](xyz, feat)
            total = total + model[1](xyz, nx, feat, nf).square().mean()
        return total

    ref, model = make(), make()
    ref_opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    opt = FlatAdam(model.parameters(), lr=1e-3)
    for it in range(3):
        ref.load_state_dict(model.state_dict())        # same weights and BN buffers going into the step
        for m, o in ((ref, ref_opt), (model, opt)):
            o.zero_grad(set_to_none=True)
            loss_of(m).backward()
        torch.cuda.synchronize()
        for (name, p), q in zip(model.named_parameters(), ref.parameters()):
            torch.testing.assert_close(p.grad, q.grad, rtol=2e-4, atol=1e-6, msg=lambda s: f"step {it} {name}: {s}")
        # Adam divides by sqrt(v): parameters whose gradient is rounding noise (a conv weight's scale under train-mode
        # BN) would amplify 1e-7 differences to lr-sized ones, so the update itself is checked on IDENTICAL gradients
        for p, q in zip(model.parameters(), ref.parameters()):
            q.grad = p.grad.detach().clone()
        ref_opt.step(); opt.step()
        for (name, p), q in zip(model.named_parameters(), ref.parameters()):
            torch.testing.assert_close(p, q, rtol=1e-5, atol=1e-7, msg=lambda s: f"step {it} {name}: {s}")

    # the whole encoder, siamese, against two separate single-use backward passes summed
    torch.manual_seed(1)
    enc = PointNet2MSG([[0.05, 0.1], [0.1, 0.2], [0.2, 0.4], [0.4, 0.8]]).cuda().train()
    pts = [torch.rand(2, 1024, 3, generator=g).cuda() - 0.5 for _ in range(2)]
    state = {k: v.clone() for k, v in enc.state_dict().items()}
    want = None
    for x in pts:
        enc.load_state_dict(state)
        enc.zero_grad(set_to_none=True)
        enc(x).square().mean().backward()
        grads = [p.grad.clone() for p in enc.parameters()]
        want = grads if want is None else [a + b for a, b in zip(want, grads)]
    enc.load_state_dict(state)
    enc_opt = FlatAdam(enc.parameters(), lr=1e-3)
    enc_opt.zero_grad(set_to_none=True)
    (enc(pts[0]).square().mean() + enc(pts[1]).square().mean()).backward()
    torch.cuda.synchronize()
    for (name, p), w in zip(enc.named_parameters(), want):
        torch.testing.assert_close(p.grad, w, rtol=2e-4, atol=1e-6, msg=lambda s: f"{name}: {s}")


def test_flat_adam_keeps_channels_last_parameters_channels_last():
    """A channels-last convolution weight stays channels-last as a view of the flat buffer (no per-call layout conversion
    on the GPU), and updates, packed gradients and the state_dict are those of the logical tensor."""
    torch.manual_seed(2)
    a = torch.nn.Sequential(torch.nn.Conv2d(4, 6, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(6, 2, 1))
    b = torch.nn.Sequential(torch.nn.Conv2d(4, 6, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(6, 2, 1))
    b.load_state_dict(a.state_dict())
    a, b = a.to(memory_format=torch.channels_last), b.to(memory_format=torch.channels_last)
    ref = torch.optim.Adam(a.parameters(), lr=1e-2)
    opt = FlatAdam(b.parameters(), lr=1e-2)
    w = b[0].weight
    assert w.is_contiguous(memory_format=torch.channels_last) and not w.is_contiguous()
    assert w.untyped_storage().data_ptr() == opt.flat.untyped_storage().data_ptr()
    torch.testing.assert_close(w, a[0].weight)
    g = torch.Generator().manual_seed(0)
    for _ in range(3):
        x = torch.randn(2, 4, 5, 5, generator=g).contiguous(memory_format=torch.channels_last)
        for m, o in ((a, ref), (b, opt)):
            o.zero_grad(set_to_none=True)
            m(x).square().mean().backward()
            o.step()
    for p, q in zip(a.parameters(), b.parameters()):
        torch.testing.assert_close(q, p, rtol=1e-5, atol=1e-7)
    sd, sd_ref = opt.state_dict(), ref.state_dict()
    for k in sd_ref["state"]:
        torch.testing.assert_close(sd["state"][k]["exp_avg"], sd_ref["state"][k]["exp_avg"], rtol=1e-5, atol=1e-8)
        torch.testing.assert_close(sd["state"][k]["exp_avg_sq"], sd_ref["state"][k]["exp_avg_sq"], rtol=1e-5, atol=1e-10)
    opt2 = FlatAdam(b.parameters(), lr=1e-2)
    opt2.load_state_dict(sd)
    torch.testing.assert_close(opt2.exp_avg, opt.exp_avg)
    for v, p in zip(opt.grad_views(opt.pack_grads()), opt.params):
        torch.testing.assert_close(v, p.grad)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [4, 1031, 1 << 20, (1 << 21) + 5])
def test_counting_adam_launch_equals_increment_then_update(n):
    """istnet_adam_step_counting: the launch that advances the step count itself gives bit for bit what `step += 1` followed by
    istnet_adam_step gives, the count goes up by one per launch (also on replays of a captured launch), and the ticket is back
    at zero afterwards -- sizes below one workgroup, odd tails, and more quads than the capped grid covers in one sweep."""
    from istnet_amd import _native
    lib = _native.lib()
    dev = "cuda:0"
    g = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=g).to(dev)
    grads = [torch.randn(n, generator=g).to(dev) for _ in range(3)]
    st = torch.cuda.current_stream().cuda_stream

    def state():
        return p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros((), device=dev)

    pa, ma, va, sa = state()
    pb, mb, vb, sb = state()
    ticket = torch.zeros((), dtype=torch.int32, device=dev)
    args = (None, 1e-3, 0.9, 0.999, 1e-8, 0.01, 0.5, st)
    for gr in grads:
        sa += 1
        assert lib.istnet_adam_step(n, pa.data_ptr(), gr.data_ptr(), ma.data_ptr(), va.data_ptr(), sa.data_ptr(), *args) == 0
        assert lib.istnet_adam_step_counting(n, pb.data_ptr(), gr.data_ptr(), mb.data_ptr(), vb.data_ptr(), sb.data_ptr(),
                                             ticket.data_ptr(), *args) == 0
    torch.cuda.synchronize()
    assert float(sb) == 3.0 and int(ticket) == 0
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph):
            assert lib.istnet_adam_step_counting(n, pb.data_ptr(), grads[0].data_ptr(), mb.data_ptr(), vb.data_ptr(),
                                                 sb.data_ptr(), ticket.data_ptr(), None, 1e-3, 0.9, 0.999, 1e-8, 0.01, 0.5,
                                                 torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(4):
        graph.replay()
    torch.cuda.synchronize()
    assert float(sb) == 7.0 and int(ticket) == 0
    for _ in range(4):
        sa += 1
        assert lib.istnet_adam_step(n, pa.data_ptr(), grads[0].data_ptr(), ma.data_ptr(), va.data_ptr(), sa.data_ptr(), *args) == 0
    torch.cuda.synchronize()
    assert torch.equal(pa, pb)
