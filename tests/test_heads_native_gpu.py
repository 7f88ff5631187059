"""Native tail of the pose estimators and the pose / NOCS losses (csrc/pose_tail.hip) against the reference composition
evaluated in float64 (model/ist_net.py:250-264,318-332, utils/rotation_utils.py:4-28, model/losses.py:3-49): outputs and
every gradient within 1e-4 relative -- no max-pool here, so no arg-max routing to excuse anything."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def _torch_path(fn):
    """Run fn with the native tail switched off (literal torch composition)."""
    from istnet_amd import heads_native
    saved = heads_native.USE_NATIVE_TAIL
    heads_native.USE_NATIVE_TAIL = False
    try:
        return fn()
    finally:
        heads_native.USE_NATIVE_TAIL = saved


@pytest.mark.parametrize("b", [1, 7, 32, 64, 200])
def test_ortho6d_forward_backward(b):
    from istnet_amd.rotation_utils import Ortho6d2Mat, ortho6d_to_mat
    g = torch.Generator().manual_seed(b)
    r6 = torch.randn(b, 6, generator=g).to(DEV)
    wgt = torch.randn(b, 3, 3, generator=g).to(DEV)
    a = r6.clone().requires_grad_(True)
    out = ortho6d_to_mat(a)
    (out * wgt).sum().backward()
    ref_in = r6.double().clone().requires_grad_(True)
    ref = _torch_path(lambda: Ortho6d2Mat(ref_in[:, :3], ref_in[:, 3:]))
    (ref * wgt.double()).sum().backward()
    assert _rel(out, ref) < 1e-6
    assert _rel(a.grad, ref_in.grad) < 1e-5
    # a rotation: orthonormal columns, determinant +1
    eye = torch.eye(3, device=DEV).expand(b, 3, 3)
    torch.testing.assert_close(out.transpose(1, 2) @ out, eye, rtol=0, atol=1e-5)
    assert float((torch.linalg.det(out) - 1).abs().max()) < 1e-5
    # the two-argument public form takes the same kernel
    b2 = r6.clone().requires_grad_(True)
    out2 = Ortho6d2Mat(b2[:, :3], b2[:, 3:])
    assert torch.equal(out2, out)


def test_pose_dis_and_smooth_l1():
    from istnet_amd import losses
    g = torch.Generator().manual_seed(3)
    b, n = 32, 1024
    r1 = torch.randn(b, 3, 3, generator=g).to(DEV); r2 = torch.randn(b, 3, 3, generator=g).to(DEV)
    t1 = torch.randn(b, 3, generator=g).to(DEV); t2 = torch.randn(b, 3, generator=g).to(DEV)
    s1 = torch.rand(b, 3, generator=g).to(DEV); s2 = torch.rand(b, 3, generator=g).to(DEV)
    ins = [t.clone().requires_grad_(True) for t in (r1, t1, s1)]
    loss = losses.PoseDis(*ins, r2, t2, s2)
    (loss * 1.7).backward()
    ins64 = [t.double().clone().requires_grad_(True) for t in (r1, t1, s1)]
    ref = _torch_path(lambda: losses.PoseDis(*ins64, r2.double(), t2.double(), s2.double()))
    (ref * 1.7).backward()
    assert _rel(loss, ref) < 1e-6
    for a, c in zip(ins, ins64):
        assert _rel(a.grad, c.grad) < 1e-5
    p1 = (torch.randn(b, n, 3, generator=g) * 0.12).to(DEV)        # both branches of the smooth-L1 (threshold 0.1)
    p2 = (torch.randn(b, n, 3, generator=g) * 0.12).to(DEV)
    a = p1.clone().requires_grad_(True)
    l1 = losses.SmoothL1Dis(a, p2)
    (l1 * 0.3).backward()
    a64 = p1.double().clone().requires_grad_(True)
    ref1 = _torch_path(lambda: losses.SmoothL1Dis(a64, p2.double()))
    (ref1 * 0.3).backward()
    assert _rel(l1, ref1) < 1e-6
    assert _rel(a.grad, a64.grad) < 1e-5


@pytest.mark.parametrize("b", [2, 5, 32, 33, 64])
def test_fc_heads_forward_backward_vs_float64(b):
    """The three heads of an estimator: outputs, the gradient of the pooled feature and of all 18 parameters within 1e-4
    relative of a float64 evaluation of the nn.Sequential modules."""
    from istnet_amd.ist_net import HeavyEstimator
    from istnet_amd import heads_native
    torch.manual_seed(b)
    est = HeavyEstimator().to(DEV)
    heads = [est.rotation_estimator, est.translation_estimator, est.size_estimator]
    g = torch.Generator().manual_seed(b + 1)
    pooled = torch.rand(b, 512, generator=g).to(DEV)                # a mean of ReLU outputs: non-negative
    wg = [torch.randn(b, k, generator=g).to(DEV) for k in (6, 3, 3)]
    x = pooled.clone().requires_grad_(True)
    outs = heads_native.fc_heads(heads, x)
    assert outs is not None
    sum((o * w).sum() for o, w in zip(outs, wg)).backward()
    got = [p.grad.clone() for h in heads for p in h.parameters()]
    heads64 = [copy.deepcopy(h).double() for h in heads]
    for h in heads64:
        h.zero_grad()
    x64 = pooled.double().clone().requires_grad_(True)
    outs64 = [h(x64) for h in heads64]
    sum((o * w.double()).sum() for o, w in zip(outs64, wg)).backward()
    for o, o64 in zip(outs, outs64):
        assert _rel(o, o64) < 1e-5
    assert _rel(x.grad, x64.grad) < 1e-4
    for gg, p64 in zip(got, [p for h in heads64 for p in h.parameters()]):
        assert _rel(gg, p64.grad) < 1e-4


def test_estimator_heads_gradients_vs_float64():
    """A whole HeavyEstimator (per-point stacks on the MFMA kernels, mean-pool, native heads, native Ortho6d2Mat) and
    PoseDis: every parameter gradient within 1e-4 relative (L2 over the tensor) of the float64 torch composition."""
    from istnet_amd.ist_net import HeavyEstimator
    from istnet_amd import losses
    torch.manual_seed(0)
    est = HeavyEstimator().to(DEV).train()
    g = torch.Generator().manual_seed(1)
    b, n = 4, 256
    pts = (torch.randn(b, n, 3, generator=g) * 0.1).to(DEV)
    pts_w = (torch.rand(b, n, 3, generator=g) - 0.5).to(DEV)
    rgb = torch.randn(b, 128, n, generator=g).to(DEV)
    loc = torch.randn(b, 128, n, generator=g).to(DEV)
    loc_w = torch.randn(b, 128, n, generator=g).to(DEV)
    r2 = torch.linalg.qr(torch.randn(b, 3, 3, generator=g))[0].to(DEV)
    t2 = torch.randn(b, 3, generator=g).to(DEV)
    s2 = torch.rand(b, 3, generator=g).to(DEV)

    def run(model, cast):
        model.zero_grad(set_to_none=True)
        r, t, s = model(cast(pts), cast(pts_w), cast(rgb), cast(loc), cast(loc_w))
        loss = losses.PoseDis(r, t, s, cast(r2), cast(t2), cast(s2))
        loss.backward()
        return loss.detach(), [p.grad.detach().clone() for p in model.parameters()]

    loss, grads = run(est, lambda t: t)
    est64 = copy.deepcopy(est).double()
    loss64, grads64 = _torch_path(lambda: run(est64, lambda t: t.double()))
    assert _rel(loss, loss64) < 1e-5
    names = [n for n, _ in est.named_parameters()]
    for name, a, c in zip(names, grads, grads64):
        err = float((a.double() - c).norm() / (c.norm() + 1e-30))
        assert err < 1e-4, (name, err)


def test_stack_with_fused_mean_pool_equals_stack_then_mean():
    """pose_mlp2 = [conv, relu, conv, relu, AdaptiveAvgPool1d(1)] on [feat, global mean of feat]: the pooled form (mean of the
    ReLU output taken from the last raw activation, dense gradient re-expanded in backward) against the stack followed by
    torch's mean -- value and every gradient."""
    from istnet_amd.pointnet2 import fused_mlp
    torch.manual_seed(4)
    b, n = 4, 512
    seq = torch.nn.Sequential(torch.nn.Conv1d(512, 512, 1), torch.nn.ReLU(), torch.nn.Conv1d(512, 512, 1), torch.nn.ReLU()).to(DEV)
    feat0 = torch.relu(torch.randn(b, 256, n, device=DEV))
    wgt = torch.randn(b, 512, device=DEV)
    res = []
    for pooled in (True, False):
        seq.zero_grad(set_to_none=True)
        feat = feat0.clone().requires_grad_(True)
        out = fused_mlp.pointwise_conv_stack_multi(seq, [feat], with_mean=True, pool_mean=pooled)
        if not pooled:
            out = out.mean(dim=2)
        (out * wgt).sum().backward()
        res.append((out.detach(), feat.grad.clone(), [p.grad.clone() for p in seq.parameters()]))
    assert res[0][0].shape == (b, 512)
    assert _rel(res[0][0], res[1][0]) < 1e-6
    assert _rel(res[0][1], res[1][1]) < 1e-5
    for a, c in zip(res[0][2], res[1][2]):
        assert _rel(a, c) < 1e-5


@pytest.mark.parametrize("n,with_target", [(32 * 128 * 1024, False), (4099, True), (7, False), (32 * 128 * 1024, True)])
def test_mse_value_and_gradient_from_one_pass(n, with_target):
    """losses.mse_value_and_grad (istnet_mse_value_grad) against nn.functional.mse_loss and its autograd gradient in float64."""
    from istnet_amd import losses
    g = torch.Generator().manual_seed(n)
    a = torch.randn(n, generator=g).to(DEV)
    b = torch.randn(n, generator=g).to(DEV) if with_target else None
    loss, grad = losses.mse_value_and_grad(a, b)
    a64 = a.double().requires_grad_(True)
    want = torch.nn.functional.mse_loss(a64, b.double() if b is not None else torch.zeros_like(a64))
    want.backward()
    assert abs(float(loss) - float(want)) <= 1e-6 * abs(float(want))
    torch.testing.assert_close(grad.double(), a64.grad, rtol=1e-6, atol=1e-12)
    # deterministic: the partial sums are added in a fixed order
    loss2, grad2 = losses.mse_value_and_grad(a, b)
    assert torch.equal(loss, loss2) and torch.equal(grad, grad2)
    # CPU tensors take the torch expression
    lc, gc = losses.mse_value_and_grad(a.cpu(), b.cpu() if b is not None else None)
    assert abs(float(lc) - float(want)) <= 1e-5 * abs(float(want))
