"""The last layer of a set-abstraction scale with the max-pool in the GEMM's epilogue (pw_fwd2_kernel<..., POOL> +
istnet_bn_finalize_pool_apply, the default since round 4).  Against the separate pool pass of the same library (forward
bit for bit; gradients to fp32 round-off -- they differ only where two slots tie after BatchNorm's rounding) and against
a float64 evaluation of the reference composition Conv2d 1x1 -> BatchNorm2d -> ReLU -> max_pool2d
(pointnet2_modules.py:61-71)."""
import copy

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture()
def small_launches():
    """Let the direct-operand forward kernel (the one with the pooled epilogue) take launches of any size, so the path runs
    at test-sized batches; restored afterwards."""
    from istnet_amd import _native
    lib = _native.lib()
    assert lib.istnet_pw_set_tuning(14, 1) == 0
    yield
    assert lib.istnet_pw_set_tuning(14, 1024) == 0


def _stack(spec, seed, gamma_signs=False):
    from istnet_amd.pointnet2.pytorch_utils import SharedMLP
    torch.manual_seed(seed)
    mlp = SharedMLP(list(spec), bn=True).to(DEV)
    gen = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for mod in mlp.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                w = torch.rand(mod.weight.shape, generator=gen) + 0.5
                if gamma_signs:      # negative and exactly-zero scales: the pooled value is then the raw MINIMUM / a constant
                    w[::3] *= -1.0
                    w[1] = 0.0
                mod.weight.copy_(w.to(DEV))
                mod.bias.copy_((torch.randn(mod.bias.shape, generator=gen) * 0.3).to(DEV))
                mod.running_mean.copy_((torch.randn(mod.bias.shape, generator=gen) * 0.1).to(DEV))
                mod.running_var.copy_((torch.rand(mod.bias.shape, generator=gen) + 0.5).to(DEV))
    return mlp


def _run(mlp, x, wgt, kind, train=True):
    from istnet_amd.pointnet2 import fused_mlp
    m = copy.deepcopy(mlp).train(train)
    xx = x.clone().requires_grad_(True)
    if kind == "f64":
        m, xx = m.double(), x.double().clone().requires_grad_(True)
        act = m(xx)
        out = F.max_pool2d(act, kernel_size=[1, act.size(3)]).squeeze(-1)
    else:
        saved = fused_mlp.USE_POOL_EPILOGUE
        before = fused_mlp.STATS["pool_epilogue"]
        try:
            fused_mlp.USE_POOL_EPILOGUE = kind == "new"
            out = fused_mlp.shared_mlp_maxpool(m, xx)
        finally:
            fused_mlp.USE_POOL_EPILOGUE = saved
        assert (fused_mlp.STATS["pool_epilogue"] - before == 1) == (kind == "new"), "wrong path taken"
    (out * wgt.to(out.dtype)).sum().backward()
    torch.cuda.synchronize()
    return (out.detach(), xx.grad.detach(), {n: p.grad.detach().clone() for n, p in m.named_parameters()},
            {n: t.detach().clone() for n, t in m.named_buffers()})


@pytest.mark.parametrize("cin,cout,s", [(32, 64, 16), (32, 64, 32), (64, 128, 16), (64, 128, 32), (128, 256, 16),
                                        (128, 256, 32), (64, 64, 32), (128, 128, 16)])
@pytest.mark.parametrize("gamma_signs", [False, True])
def test_last_layer_pooled_epilogue(small_launches, cin, cout, s, gamma_signs):
    b, g = 3, 2048 // s
    spec = [19, cin, cout]
    mlp = _stack(spec, seed=cin + cout + s, gamma_signs=gamma_signs)
    gen = torch.Generator().manual_seed(5)
    x = (torch.randn(b, spec[0], g, s, generator=gen) * 0.7 + 0.1).to(DEV)
    wgt = torch.randn(b, cout, g, generator=gen).to(DEV)
    o_new, dx_new, g_new, st_new = _run(mlp, x, wgt, "new")
    o_old, dx_old, g_old, st_old = _run(mlp, x, wgt, "old")
    o_64, dx_64, g_64, _ = _run(mlp, x, wgt, "f64")
    # forward: the same statistics partials, the same affine map applied to the same raw extremum -> bit for bit
    assert torch.equal(o_new, o_old)
    for n in st_old:
        assert torch.equal(st_new[n], st_old[n]), n
    assert float((o_new.double() - o_64).abs().max() / o_64.abs().max()) < 2e-5

    def close(name, got, ref64, other):
        scale = ref64.abs().max() + 1e-30
        err = float((got.double() - ref64).abs().max() / scale)
        err_other = float((other.double() - ref64).abs().max() / scale)
        # as close to float64 as the stored-activation path is (twice its error + fp32 round-off), and within 1e-4 outright
        assert err < 1e-4 and err <= 2.0 * err_other + 2e-6, (name, err, err_other)
    close("dx", dx_new, dx_64, dx_old)
    for n in g_64:
        close(n, g_new[n], g_64[n], g_old[n])


def test_last_layer_pooled_epilogue_eval_mode(small_launches):
    """Fixed (running-statistics) BatchNorm: forward equals the separate pool pass bit for bit (istnet_pool_apply)."""
    spec, b, g, s = [35, 64, 128], 2, 128, 16
    mlp = _stack(spec, seed=3)
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(b, spec[0], g, s, generator=gen).to(DEV)
    wgt = torch.randn(b, spec[-1], g, generator=gen).to(DEV)
    o_new, dx_new, g_new, _ = _run(mlp, x, wgt, "new", train=False)
    o_old, dx_old, g_old, _ = _run(mlp, x, wgt, "old", train=False)
    assert torch.equal(o_new, o_old)
    torch.testing.assert_close(dx_new, dx_old, rtol=1e-4, atol=1e-5)
    for n in g_old:
        torch.testing.assert_close(g_new[n], g_old[n], rtol=1e-4, atol=1e-4 * float(g_old[n].abs().max()))
