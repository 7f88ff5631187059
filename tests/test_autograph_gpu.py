"""graphed.AutoGraph: the eager caller's loop (utils/solver.py:88-99) replays captured HIP graphs -- same kernels, same
order, so every output and gradient must be BIT-identical with the launch-by-launch path, and every situation in which a
replay would be wrong must take the plain path."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CAM_RADII = [[0.01, 0.02], [0.02, 0.04], [0.04, 0.08], [0.08, 0.16]]


def _cloud(b, n, seed):
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(b, n, 3, generator=g)
    return (d / d.norm(dim=2, keepdim=True) * 0.1 + torch.randn(b, n, 3, generator=g) * 0.002).cuda().contiguous()


def _model(seed=0):
    from istnet_amd.modules import PointNet2MSG
    torch.manual_seed(seed)
    return PointNet2MSG([list(r) for r in CAM_RADII]).cuda().train()


def _train(auto, opt_kind, steps=7, b=4, n=512):
    """Reference-style loop; returns per-step (loss, output) and the final parameters / running statistics."""
    from istnet_amd import graphed
    from istnet_amd.optim import FlatAdam, layout_hints
    graphed.ENABLED = auto
    try:
        model = _model()
        opt = (torch.optim.Adam(model.parameters(), lr=1e-3) if opt_kind == "torch"
               else FlatAdam(model.parameters(), lr=1e-3, adjacent=layout_hints(model)))
        outs = []
        for it in range(steps):
            pts = _cloud(b, n, 100 + it)
            opt.zero_grad()
            out = model(pts)
            loss = out.square().mean()
            loss.backward()
            opt.step()
            outs.append((loss.detach().clone(), out.detach()))      # outputs are held across steps on purpose
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        return outs, state
    finally:
        graphed.ENABLED = True


@pytest.mark.parametrize("opt_kind", ["torch", "flat"])
def test_graphed_training_loop_is_bit_identical_with_the_plain_path(opt_kind):
    from istnet_amd import graphed
    before = dict(graphed.STATS)
    graphed.WHY.clear()
    outs_g, state_g = _train(True, opt_kind)
    assert graphed.STATS["captures"] == before["captures"] + 1, dict(graphed.WHY)
    assert graphed.STATS["replays"] >= before["replays"] + 5          # steps 3..7 replayed
    assert graphed.STATS["failed"] == before["failed"]
    outs_p, state_p = _train(False, opt_kind)
    for (lg, og), (lp, op) in zip(outs_g, outs_p):
        assert torch.equal(lg, lp)
        assert torch.equal(og, op)
    assert state_g.keys() == state_p.keys()
    for k in state_g:
        assert torch.equal(state_g[k], state_p[k]), k


def test_two_forwards_before_backward_take_the_plain_path():
    from istnet_amd import graphed
    model = _model()
    a, b = _cloud(2, 256, 1), _cloud(2, 256, 2)
    for _ in range(3):                       # warm-up + capture + one replay
        model.zero_grad()
        model(a).square().mean().backward()
    replays = graphed.STATS["replays"]
    model.zero_grad()
    oa = model(a)                            # graphed
    ob = model(b)                            # previous output alive and not back-propagated: plain path
    assert graphed.STATS["replays"] == replays + 1
    (oa.square().mean() + ob.square().mean()).backward()
    got = [p.grad.clone() for p in model.parameters()]
    graphed.ENABLED = False
    try:
        # BatchNorm running statistics differ by now, gradients do not depend on them in train mode
        model.zero_grad()
        (model(a).square().mean() + model(b).square().mean()).backward()
    finally:
        graphed.ENABLED = True
    for g, p in zip(got, model.parameters()):
        assert torch.equal(g, p.grad)


def test_gradient_accumulation_and_no_grad_take_the_plain_path():
    from istnet_amd import graphed
    model = _model()
    a = _cloud(2, 256, 3)
    for _ in range(3):
        model.zero_grad()
        model(a).square().mean().backward()
    replays = graphed.STATS["replays"]
    model(a).square().mean().backward()      # .grad exists: accumulate, launch by launch
    assert graphed.STATS["replays"] == replays
    with torch.no_grad():
        model(a)
    assert graphed.STATS["replays"] == replays
    g1 = [p.grad.clone() for p in model.parameters()]
    model.zero_grad()
    model(a).square().mean().backward()      # graphed again
    assert graphed.STATS["replays"] == replays + 1
    for acc, p in zip(g1, model.parameters()):
        torch.testing.assert_close(acc, 2 * p.grad, rtol=1e-6, atol=1e-9)


def test_new_shape_gets_its_own_graph_and_eval_mode_its_own_key():
    from istnet_amd import graphed
    model = _model()
    caps = graphed.STATS["captures"]
    for shape_seed, (b, n) in enumerate([(2, 256), (3, 512), (2, 256)]):
        x = _cloud(b, n, shape_seed)
        for _ in range(3):
            model.zero_grad()
            model(x).square().mean().backward()
    assert graphed.STATS["captures"] == caps + 2
    model.eval()
    x = _cloud(2, 256, 9)
    ref = None
    for it in range(4):                      # eval-mode BatchNorm with autograd on (fine-tuning): its own entry
        model.zero_grad()
        out = model(x)
        out.square().mean().backward()
        ref = out.detach().clone() if ref is None else ref
        assert torch.equal(out, ref)
    assert graphed.STATS["captures"] == caps + 3


# ---- graphed.InferenceGraph: the reference's test loop (one image per step, its instances the batch; utils/solver.py:199-262) ----

def _infer_net():
    import bench
    dev = torch.device("cuda:0")
    net = bench.make_istnet(dev, seed=0)
    g = torch.Generator().manual_seed(5)
    for m in net.modules():          # non-trivial running statistics, as after training
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
    return net.eval(), dev


def _plain(net, batch):
    from istnet_amd import graphed
    graphed.ENABLED = False
    try:
        with torch.no_grad():
            return {k: v.clone() for k, v in net(batch).items()}
    finally:
        graphed.ENABLED = True


def _close(got, ref, tol=2e-5):
    assert got.keys() == ref.keys()
    for k in ref:
        assert float((got[k] - ref[k]).abs().max()) <= tol * (1.0 + float(ref[k].abs().max())), k


def test_inference_graphs_follow_the_batch_size_and_match_the_plain_path():
    import bench
    from istnet_amd import graphed
    net, dev = _infer_net()
    before = dict(graphed.STATS)
    sizes = [1, 3, 1, 3, 5, 1]
    held = []
    for rnd in range(4):                              # two warm-up calls per size, then capture, then replays
        for i, b in enumerate(sizes):
            batch = bench.istnet_batch(b, 1024, seed=10 * rnd + i, device=dev)      # new data every call
            with torch.no_grad():
                got = net(batch)
            _close(got, _plain(net, batch))
            held.append(got["pred_rotation"])
    assert graphed.STATS["infer_captures"] == before["infer_captures"] + 3           # one per distinct batch size
    assert graphed.STATS["infer_failed"] == before["infer_failed"]
    assert graphed.STATS["infer_replays"] >= before["infer_replays"] + 9
    # the caller owns what it was handed: no two results share storage (a replay overwrites only the graph's own buffers)
    assert len({t.data_ptr() for t in held}) == len(held)
    # weights loaded IN PLACE (load_state_dict, checkpoint averaging) are what the next replay reads
    batch = bench.istnet_batch(3, 1024, seed=99, device=dev)
    with torch.no_grad():
        old = net(batch)
        for p in net.main_estimator.parameters():
            p.mul_(1.05)
        new = net(batch)
    assert graphed.STATS["infer_captures"] == before["infer_captures"] + 3
    _close(new, _plain(net, batch))
    assert float((new["pred_size"] - old["pred_size"]).abs().max()) > 1e-6
    graphed.reset(net)


def test_inference_graph_leaves_training_and_grad_mode_alone():
    import bench
    from istnet_amd import graphed
    net, dev = _infer_net()
    batch = bench.istnet_batch(2, 1024, seed=3, device=dev)
    before = dict(graphed.STATS)
    for _ in range(4):                 # eval mode WITH autograd (e.g. test-time refinement of an input): plain path
        out = net(batch)
    assert out["pred_rotation"].requires_grad
    out["pred_rotation"].sum().backward()
    assert graphed.STATS["infer_captures"] == before["infer_captures"]
    # the switch state is part of the key: flipping a library switch starts a new entry instead of replaying the old kernels
    from istnet_amd import ist_net
    with torch.no_grad():
        for _ in range(4):
            a = net(batch)
        assert graphed.STATS["infer_captures"] == before["infer_captures"] + 1
        ist_net.USE_GATHER_FIRST = False
        try:
            for _ in range(4):
                b_ = net(batch)
        finally:
            ist_net.USE_GATHER_FIRST = True
        assert graphed.STATS["infer_captures"] == before["infer_captures"] + 2
    _close(b_, a, tol=1e-4)
    graphed.reset(net)


def test_two_uses_in_one_backward_are_reproducible_without_an_optimizer_slot():
    """loss = f(m(a)) + f(m(b)) with plain parameters (no FlatAdam slot): the second producer of a weight gradient must not
    defer its launches behind the engine's sum of the two tensors (round 5: it did, and on this small shape one FP weight
    gradient changed bits from run to run).  Same step 30 times, allocator moved around in between: identical bits."""
    from istnet_amd import graphed
    graphed.ENABLED = False
    try:
        model = _model()
        a, b = _cloud(2, 256, 1), _cloud(2, 256, 2)
        ref = None
        for it in range(30):
            model.zero_grad()
            junk = torch.empty(1 << (10 + it % 12), device="cuda").normal_()
            (model(a).square().mean() + model(b).square().mean()).backward()
            got = [p.grad.clone() for p in model.parameters()]
            ref = ref or got
            for (name, _), g, r in zip(model.named_parameters(), got, ref):
                assert torch.equal(g, r), (it, name)
            del junk
    finally:
        graphed.ENABLED = True


# ---- round 6: host state a replay must follow, and the hazards of an autograd node that outlives its output ----

def _momentum_run(auto, sched, lr=1e-3, b=4, n=512):
    """Reference-style loop (utils/solver.py:88-99) in which the momentum is set the way the reference's OWN scheduler sets it
    (utils/scheduler.py BNMomentumScheduler: a bare ``m.momentum = x`` under ``model.apply``) -- no package scheduler, no
    sync_bn_momentum.  Returns the running statistics after the run."""
    from istnet_amd import graphed
    graphed.ENABLED = auto
    try:
        model = _model()
        opt = torch.optim.Adam(model.parameters(), lr=lr)
        for it, mom in enumerate(sched):
            model.apply(lambda m, mom=mom: setattr(m, "momentum", mom) if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) else None)
            pts = _cloud(b, n, 300 + it)
            opt.zero_grad()
            model(pts).square().mean().backward()
            opt.step()
        torch.cuda.synchronize()
        return {k: v.detach().clone() for k, v in model.state_dict().items()
                if k.endswith(("running_mean", "running_var", "num_batches_tracked"))}
    finally:
        graphed.ENABLED = True


def test_graph_segments_follow_a_foreign_momentum_scheduler():
    """replays read the momentum from a device slot; a caller that changes ``bn.momentum`` by hand (the
    reference's own BNMomentumScheduler) must be followed, bit for bit, and the stale value must be visibly different."""
    from istnet_amd import graphed
    sched = [0.5, 0.5, 0.5, 0.2, 0.05, 0.9, 0.9, 0.3]            # steps 1-2 warm up, 3 captures, 4.. replay with new momenta
    before = dict(graphed.STATS)
    got = _momentum_run(True, sched)
    assert graphed.STATS["captures"] == before["captures"] + 1, dict(graphed.WHY)
    assert graphed.STATS["replays"] >= before["replays"] + 6
    assert graphed.STATS["momentum_syncs"] >= before["momentum_syncs"] + 4      # 0.2, 0.05, 0.9, 0.3
    want = _momentum_run(False, sched)
    stale = _momentum_run(False, [0.5] * len(sched))              # what a momentum frozen at capture time would give
    assert len(want) >= 96
    worst = 0.0
    for k in want:
        assert torch.equal(got[k], want[k]), k
        if not k.endswith("num_batches_tracked"):
            worst = max(worst, ((stale[k] - want[k]).abs().max() / want[k].abs().max().clamp_min(1e-12)).item())
    assert worst > 1e-2


def test_eps_is_part_of_the_key():
    from istnet_amd import graphed
    model = _model()
    a = _cloud(2, 256, 5)
    for _ in range(4):
        model.zero_grad()
        model(a).square().mean().backward()
    caps, replays = graphed.STATS["captures"], graphed.STATS["replays"]
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps = 1e-3
    model.zero_grad()
    out = model(a)                              # new key: plain (warm-up), never a replay of the eps=1e-5 graph
    assert graphed.STATS["replays"] == replays and graphed.STATS["captures"] == caps
    graphed.ENABLED = False
    try:
        ref = model(a)
    finally:
        graphed.ENABLED = True
    assert torch.equal(out, ref)


def test_node_that_outlives_its_output_keeps_the_plain_path():
    """``enc(a).mean() + enc(b).mean()`` -- mean() does not save its input, so the first output tensor
    dies while its autograd node is still in the graph.  The second call must NOT replay over the first call's activations."""
    from istnet_amd import graphed
    model = _model()
    a, b = _cloud(2, 256, 1), _cloud(2, 256, 2)
    for _ in range(3):
        model.zero_grad()
        model(a).mean().backward()
    replays = graphed.STATS["replays"]
    model.zero_grad()
    loss = model(a).mean() + model(b).mean()      # first output is garbage by the time the second call runs
    assert graphed.STATS["replays"] == replays + 1, dict(graphed.WHY)
    loss.backward()
    got = [p.grad.clone() for p in model.parameters()]
    graphed.ENABLED = False
    try:
        model.zero_grad()
        (model(a).mean() + model(b).mean()).backward()
    finally:
        graphed.ENABLED = True
    for g, p in zip(got, model.parameters()):
        assert torch.equal(g, p.grad)


def test_second_backward_through_a_segment_raises():
    from istnet_amd import graphed
    model = _model()
    a = _cloud(2, 256, 4)
    for _ in range(3):
        model.zero_grad()
        model(a).square().mean().backward()
    model.zero_grad()
    replays = graphed.STATS["replays"]
    loss = model(a).square().mean()
    assert graphed.STATS["replays"] == replays + 1
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="second backward"):
        loss.backward()


def test_backward_after_optimizer_step_raises_on_the_graph_path_too():
    """forward -> optimizer.step() -> backward: the parameters the captured backward would read are no longer the ones
    the forward used.  The plain path raises (saved tensors' version counters); the replay node saves the parameters so
    that it does as well."""
    from istnet_amd import graphed
    model = _model()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    a = _cloud(2, 256, 6)
    for _ in range(3):
        opt.zero_grad()
        model(a).square().mean().backward()
        opt.step()
    opt.zero_grad()
    replays = graphed.STATS["replays"]
    loss = model(a).square().mean()
    assert graphed.STATS["replays"] == replays + 1
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    opt.step()                                   # in-place update between forward and backward
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        loss.backward()


def test_eviction_destroys_entries_and_a_stale_node_says_so():
    from istnet_amd import graphed
    model = _model()
    shapes = [(2, 256), (3, 256), (2, 512)]
    assert graphed.MAX_ENTRIES == 2
    held = None
    for i, (b, n) in enumerate(shapes):
        x = _cloud(b, n, 20 + i)
        for _ in range(3):
            model.zero_grad()
            out = model(x)
            if i == 0:
                held = out.square().mean()          # node of the first shape's last replay, never back-propagated
            else:
                out.square().mean().backward()
    ag = graphed._REGISTRY[model]
    assert len(ag.entries) == 2                  # the first shape was evicted, its graphs destroyed on this thread
    with pytest.raises(RuntimeError, match="destroyed"):
        held.backward()
    graphed.reset(model)
    assert len(ag.entries) == 0


def test_set_to_none_false_callers_are_told_once():
    from istnet_amd import graphed
    model = _model()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    a = _cloud(2, 256, 7)
    with pytest.warns(RuntimeWarning, match="set_to_none"):
        for _ in range(graphed.PLAIN_STREAK_WARN + 3):
            opt.zero_grad(set_to_none=False)
            model(a).square().mean().backward()
            opt.step()


def test_inference_graph_notices_a_hook_registered_later():
    """a forward hook registered on a SUBMODULE after the first eval calls must keep firing."""
    from istnet_amd import graphed
    import bench
    net, dev = _infer_net()
    batch = bench.istnet_batch(2, 1024, seed=4, device=dev)
    with torch.no_grad():
        for _ in range(4):
            net(batch)
    replays = graphed.STATS["infer_replays"]
    fired = []
    sub = net.main_estimator                    # a submodule the model CALLS (fused stacks read their layers' weights directly)
    h = sub.register_forward_hook(lambda m, i, o: fired.append(1))
    with torch.no_grad():
        net(batch)
    assert fired and graphed.STATS["infer_replays"] == replays
    h.remove()
    with torch.no_grad():
        net(batch)
    assert graphed.STATS["infer_replays"] == replays + 1
    graphed.reset(net)
