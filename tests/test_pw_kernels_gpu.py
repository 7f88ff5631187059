"""Direct checks of the newer C-ABI entry points of include/istnet_pw.h against plain torch expressions
(the module-level tests cover them only through the fused autograd nodes)."""
import pytest
import torch

import istnet_amd  # noqa: F401
from istnet_amd import _native

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _st():
    return torch.cuda.current_stream().cuda_stream


def _bn_block(c, g):
    return torch.stack([torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1, torch.zeros(c),
                        torch.ones(c)]).contiguous().to(DEV)


@pytest.mark.parametrize("b,n,npoint,s,cfeat,cout", [(2, 128, 32, 16, 16, 48), (3, 256, 64, 32, 0, 16), (2, 512, 128, 8, 64, 32)])
def test_gather_add_equals_grouped_product(b, n, npoint, s, cfeat, cout):
    lib = _native.lib()
    g = torch.Generator().manual_seed(b + n)
    xyz = torch.rand(b, n, 3, generator=g).to(DEV)
    new_xyz = xyz[:, :npoint].contiguous()
    idx = torch.randint(0, n, (b, npoint, s), generator=g, dtype=torch.int32).to(DEV)
    w0 = (torch.randn(cout, 3 + cfeat, generator=g) * 0.3).to(DEV)
    feat = torch.randn(b, cfeat, n, generator=g).to(DEV) if cfeat else None
    p = npoint * s
    z = None
    if cfeat:
        z = torch.empty(b, cout, n, device=DEV)
        assert lib.istnet_pw_forward_ld(b, cfeat, cout, n, feat.data_ptr(), w0.data_ptr() + 12, 3 + cfeat, None, None,
                                        z.data_ptr(), None, None, _st()) == 0
        torch.testing.assert_close(z, torch.matmul(w0[:, 3:], feat), rtol=1e-4, atol=1e-4)
    y = torch.empty(b, cout, p, device=DEV)
    nt = lib.istnet_pw_gather_add_tiles(b, p)
    part = torch.empty(2, cout, nt, device=DEV)
    assert lib.istnet_pw_gather_add(b, n, npoint, s, cout, xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(),
                                    z.data_ptr() if z is not None else None, w0.data_ptr(), 3 + cfeat, y.data_ptr(),
                                    part[0].data_ptr(), part[1].data_ptr(), _st()) == 0
    flat = idx.long().reshape(b, p)
    xrel = torch.gather(xyz, 1, flat.unsqueeze(-1).expand(-1, -1, 3)) - new_xyz.repeat_interleave(s, dim=1)
    grouped = xrel.transpose(1, 2)
    if cfeat:
        grouped = torch.cat([grouped, torch.gather(feat, 2, flat.unsqueeze(1).expand(-1, cfeat, -1))], dim=1)
    want = torch.matmul(w0, grouped)
    torch.testing.assert_close(y, want, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(part[0].sum(-1), want.sum(dim=(0, 2)), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(part[1].sum(-1), want.square().sum(dim=(0, 2)), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("cin,cout,p,pooled", [(16, 32, 512, False), (32, 32, 256, True), (16, 16, 1024, True)])
def test_bwd_small_equals_dgrad_plus_wgrad(cin, cout, p, pooled):
    lib = _native.lib()
    b, s = 3, 16
    g = torch.Generator().manual_seed(cin + cout + p)
    x = torch.randn(b, cin, p, generator=g).to(DEV)
    y = torch.randn(b, cout, p, generator=g).to(DEV)
    w = (torch.randn(cout, cin, generator=g) * 0.2).to(DEV)
    bn, bn_in = _bn_block(cout, g), _bn_block(cin, g)
    bwdc = torch.stack([torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.01,
                        torch.randn(cout, generator=g) * 0.01]).contiguous().to(DEV)
    if pooled:
        gr = p // s
        dpool = torch.randn(b, cout, gr, generator=g).to(DEV)
        arg = torch.randint(0, s, (b, cout, gr), generator=g, dtype=torch.uint8).to(DEV)
        src = (None, dpool.data_ptr(), arg.data_ptr(), s)
        gdense = torch.zeros(b, cout, gr, s, device=DEV).scatter_(3, arg.long().unsqueeze(-1), dpool.unsqueeze(-1)).reshape(b, cout, p)
    else:
        gdense = torch.randn(b, cout, p, generator=g).to(DEV)
        src = (gdense.data_ptr(), None, None, 0)
    dx = torch.empty(b, cin, p, device=DEV)
    splits = lib.istnet_pw_bwd_small_splits(b, p)
    part = torch.empty(2, cin, splits, device=DEV)
    ws = torch.empty(splits, cout, cin, device=DEV)
    assert lib.istnet_pw_bwd_small(b, cin, cout, p, src[3], w.data_ptr(), x.data_ptr(), bn_in.data_ptr(), y.data_ptr(),
                                   src[0], src[1], 0, src[2], bn.data_ptr(), bwdc.data_ptr(), dx.data_ptr(),
                                   part[0].data_ptr(), part[1].data_ptr(), ws.data_ptr(), _st()) == 0
    mask = (y * bn[0].view(1, -1, 1) + bn[1].view(1, -1, 1)) > 0
    dy = bwdc[0].view(1, -1, 1) * (gdense * mask) + bwdc[1].view(1, -1, 1) + bwdc[2].view(1, -1, 1) * y
    want_dx = torch.matmul(w.t(), dy)
    act = torch.relu(x * bn_in[0].view(1, -1, 1) + bn_in[1].view(1, -1, 1))
    want_dw = torch.einsum("bop,bip->oi", dy, act)
    gq = want_dx * (act > 0)
    torch.testing.assert_close(dx, want_dx, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(ws.sum(0), want_dw, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(part[0].sum(-1), gq.sum(dim=(0, 2)), rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(part[1].sum(-1), (gq * x).sum(dim=(0, 2)), rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize("cin,cout,p,pooled", [(32, 64, 1024, True), (64, 64, 512, False), (64, 128, 2048, True),
                                               (128, 128, 256, False), (128, 128, 1024, True), (32, 64, 384, False)])
def test_bwd_mid_equals_dgrad_plus_wgrad(cin, cout, p, pooled):
    """pw_bwd_mid_kernel (64 / 128-channel layers) against float64 expressions of dA, dW and the statistics sums."""
    lib = _native.lib()
    b, s = 3, 16
    assert lib.istnet_pw_bwd_mid_ok(cin, cout, p)
    assert not lib.istnet_pw_bwd_mid_ok(cin, cout, p + 32) and not lib.istnet_pw_bwd_mid_ok(96, cout, p)
    g = torch.Generator().manual_seed(cin + cout + p)
    x = torch.randn(b, cin, p, generator=g).to(DEV)
    y = torch.randn(b, cout, p, generator=g).to(DEV)
    w = (torch.randn(cout, cin, generator=g) * 0.2).to(DEV)
    bn, bn_in = _bn_block(cout, g), _bn_block(cin, g)
    bwdc = torch.stack([torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.01,
                        torch.randn(cout, generator=g) * 0.01]).contiguous().to(DEV)
    if pooled:
        gr = p // s
        dpool = torch.randn(b, cout, gr, generator=g).to(DEV)
        arg = torch.randint(0, s, (b, cout, gr), generator=g, dtype=torch.uint8).to(DEV)
        src = (None, dpool.data_ptr(), arg.data_ptr(), s)
        gdense = torch.zeros(b, cout, gr, s, device=DEV).scatter_(3, arg.long().unsqueeze(-1), dpool.unsqueeze(-1)).reshape(b, cout, p)
    else:
        gdense = torch.randn(b, cout, p, generator=g).to(DEV)
        src = (gdense.data_ptr(), None, None, 0)
    dx = torch.full((b, cin, p), float("nan"), device=DEV)
    splits = lib.istnet_pw_bwd_mid_splits(b, cin, cout, p)
    assert splits >= 1
    part = torch.full((2, cin, splits), float("nan"), device=DEV)
    ws = torch.full((splits, cout, cin), float("nan"), device=DEV)
    assert lib.istnet_pw_bwd_mid(b, cin, cout, p, src[3], w.data_ptr(), x.data_ptr(), bn_in.data_ptr(), y.data_ptr(),
                                 src[0], src[1], 0, src[2], bn.data_ptr(), bwdc.data_ptr(), dx.data_ptr(),
                                 part[0].data_ptr(), part[1].data_ptr(), ws.data_ptr(), _st()) == 0
    d = torch.float64
    mask = (y * bn[0].view(1, -1, 1) + bn[1].view(1, -1, 1)) > 0          # fp32, as the kernel decides it
    dy = bwdc[0].to(d).view(1, -1, 1) * (gdense.to(d) * mask) + bwdc[1].to(d).view(1, -1, 1) + bwdc[2].to(d).view(1, -1, 1) * y.to(d)
    want_dx = torch.matmul(w.to(d).t(), dy)
    pre = x * bn_in[0].view(1, -1, 1) + bn_in[1].view(1, -1, 1)
    act = torch.relu(pre).to(d)
    want_dw = torch.einsum("bop,bip->oi", dy, act)
    gq = want_dx * (pre > 0)
    torch.testing.assert_close(dx.to(d), want_dx, rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(ws.to(d).sum(0), want_dw, rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(part[0].to(d).sum(-1), gq.sum(dim=(0, 2)), rtol=1e-4, atol=2e-3)
    torch.testing.assert_close(part[1].to(d).sum(-1), (gq * x.to(d)).sum(dim=(0, 2)), rtol=1e-4, atol=2e-3)
    # the workgroup count is a tuning parameter: another split of the points gives the same sums
    assert lib.istnet_pw_set_tuning(8, 7) == 0
    try:
        s2 = lib.istnet_pw_bwd_mid_splits(b, cin, cout, p)
        assert 1 <= s2 <= 7
        dx2, part2, ws2 = torch.empty_like(dx), torch.empty(2, cin, s2, device=DEV), torch.empty(s2, cout, cin, device=DEV)
        assert lib.istnet_pw_bwd_mid(b, cin, cout, p, src[3], w.data_ptr(), x.data_ptr(), bn_in.data_ptr(), y.data_ptr(),
                                     src[0], src[1], 0, src[2], bn.data_ptr(), bwdc.data_ptr(), dx2.data_ptr(),
                                     part2[0].data_ptr(), part2[1].data_ptr(), ws2.data_ptr(), _st()) == 0
    finally:
        lib.istnet_pw_set_tuning(8, 0)
    assert torch.equal(dx2, dx)
    torch.testing.assert_close(ws2.to(d).sum(0), want_dw, rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(part2[0].to(d).sum(-1), gq.sum(dim=(0, 2)), rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("cin,cout,p,pooled,has_bn", [(320, 256, 512, False, True), (128, 256, 1024, True, True),
                                                      (64, 64, 256, False, False), (768, 512, 128, False, True),
                                                      (96, 160, 384, True, True)])
def test_wgrad_role_split_kernel(cin, cout, p, pooled, has_bn):
    """pw_wgrad2_kernel (dense input, cin and cout >= 64): split-K partials against a float64 product, with partial
    tiles, both gradient sources, with / without an input BN block; the old kernel (tuning key 11 = 0) gives the same
    weight gradient."""
    lib = _native.lib()
    b, s = 3, 16
    g = torch.Generator().manual_seed(cin + cout + p)
    x = torch.randn(b, cin, p, generator=g).to(DEV)
    y = torch.randn(b, cout, p, generator=g).to(DEV)
    bn, bn_in = _bn_block(cout, g), _bn_block(cin, g)
    bwdc = torch.stack([torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.01,
                        torch.randn(cout, generator=g) * 0.01]).contiguous().to(DEV)
    if pooled:
        gr = p // s
        dpool = torch.randn(b, cout, gr, generator=g).to(DEV)
        arg = torch.randint(0, s, (b, cout, gr), generator=g, dtype=torch.uint8).to(DEV)
        src = (None, dpool.data_ptr(), 0, arg.data_ptr())
        gdense = torch.zeros(b, cout, gr, s, device=DEV).scatter_(3, arg.long().unsqueeze(-1), dpool.unsqueeze(-1)).reshape(b, cout, p)
    else:
        gdense = torch.randn(b, cout, p, generator=g).to(DEV)
        src = (gdense.data_ptr(), None, 0, None)
    d = torch.float64
    mask = (y * bn[0].view(1, -1, 1) + bn[1].view(1, -1, 1)) > 0
    dy = bwdc[0].to(d).view(1, -1, 1) * (gdense.to(d) * mask) + bwdc[1].to(d).view(1, -1, 1) + bwdc[2].to(d).view(1, -1, 1) * y.to(d)
    act = torch.relu(x * bn_in[0].view(1, -1, 1) + bn_in[1].view(1, -1, 1)).to(d) if has_bn else x.to(d)
    want = torch.einsum("bop,bip->oi", dy, act)
    sc, sh = (bn_in[0].data_ptr(), bn_in[1].data_ptr()) if has_bn else (None, None)
    got = []
    for enable in (1, 0):
        assert lib.istnet_pw_set_tuning(11, enable) == 0
        try:
            splits = lib.istnet_pw_wgrad_splits(b, cin, cout, p)
            ws = torch.full((splits, cout, cin), float("nan"), device=DEV)
            assert lib.istnet_pw_wgrad(b, cin, cout, p, s if pooled else 0, x.data_ptr(), sc, sh, y.data_ptr(), *src,
                                       bn.data_ptr(), bwdc.data_ptr(), ws.data_ptr(), _st()) == 0
            got.append(ws.to(d).sum(0))
        finally:
            lib.istnet_pw_set_tuning(11, 1)
    tol = dict(rtol=1e-5, atol=2e-5 * (b * p) ** 0.5)
    torch.testing.assert_close(got[0], want, **tol)
    torch.testing.assert_close(got[1], want, **tol)


@pytest.mark.parametrize("b,cin,cout,p,has_bn", [(3, 64, 128, 512, True), (2, 16, 16, 1024, True), (2, 32, 64, 384, False),
                                                 (4, 768, 512, 128, True), (2, 48, 96, 256, True), (3, 128, 40, 640, True),
                                                 (32, 256, 256, 256, True)])
def test_forward_direct_operand_kernel(b, cin, cout, p, has_bn):
    """pw_fwd2_kernel (activation operand from global memory straight into the MFMA) against a float64 product: output,
    statistics partials, partial row tiles (cout = 40, 96), every workgroup shape; the LDS-tiled kernel (tuning key
    13 = 0) agrees."""
    lib = _native.lib()
    assert lib.istnet_pw_set_tuning(14, 1) == 0      # (the launch-size threshold would send these small cases to pw_fwd_kernel)
    assert lib.istnet_pw_forward_cfg(b, cin, cout, p) > 0
    assert lib.istnet_pw_forward_cfg(b, cin + 8, cout, p) in (0, 1) and lib.istnet_pw_forward_cfg(b, cin, cout, p + 4) == 0   # (1: the split-K kernel)
    g = torch.Generator().manual_seed(b + cin + cout + p)
    x = torch.randn(b, cin, p, generator=g).to(DEV)
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).to(DEV)
    bn_in = _bn_block(cin, g)
    sc, sh = (bn_in[0].data_ptr(), bn_in[1].data_ptr()) if has_bn else (None, None)
    d = torch.float64
    act = torch.relu(x * bn_in[0].view(1, -1, 1) + bn_in[1].view(1, -1, 1)).to(d) if has_bn else x.to(d)
    want = torch.matmul(w.to(d), act)
    outs = []
    for enable in (1, 0):
        assert lib.istnet_pw_set_tuning(13, enable) == 0
        try:
            nt = lib.istnet_pw_forward_tiles(b, cin, cout, p)
            y = torch.full((b, cout, p), float("nan"), device=DEV)
            part = torch.full((2, cout, nt), float("nan"), device=DEV)
            assert lib.istnet_pw_forward(b, cin, cout, p, x.data_ptr(), w.data_ptr(), sc, sh, y.data_ptr(),
                                         part[0].data_ptr(), part[1].data_ptr(), _st()) == 0
            outs.append((y, part))
        finally:
            lib.istnet_pw_set_tuning(13, 1)
    for y, part in outs:
        torch.testing.assert_close(y.to(d), want, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(part[0].to(d).sum(-1), want.sum(dim=(0, 2)), rtol=1e-5, atol=1e-3)
        torch.testing.assert_close(part[1].to(d).sum(-1), want.square().sum(dim=(0, 2)), rtol=1e-5, atol=1e-3)
    # without statistics (eval mode)
    y2 = torch.empty(b, cout, p, device=DEV)
    assert lib.istnet_pw_forward(b, cin, cout, p, x.data_ptr(), w.data_ptr(), sc, sh, y2.data_ptr(), None, None, _st()) == 0
    assert torch.equal(y2, outs[0][0])
    assert lib.istnet_pw_set_tuning(14, 0) == 0
    assert lib.istnet_pw_forward_cfg(2, 64, 64, 1024) in (0, 1) and lib.istnet_pw_forward_cfg(32, 64, 128, 4096) > 1


@pytest.mark.parametrize("b,cin,cout,p,has_bn,mode", [(4, 512, 512, 128, True, "plain"), (2, 256, 96, 256, False, "plain"),
                                                      (3, 64, 40, 384, True, "plain"), (4, 256, 512, 128, False, "acc"),
                                                      (2, 128, 256, 256, False, "ld"), (2, 72, 64, 128, False, "ld_unaligned"),
                                                      # tiles that span several small clouds (the coarsest propagation level)
                                                      (32, 512, 512, 64, True, "ld"), (8, 128, 96, 32, True, "plain"),
                                                      (6, 64, 64, 64, False, "acc")])
def test_forward_split_k_kernel_for_small_launches(b, cin, cout, p, has_bn, mode):
    """pw_fwd_sk_kernel (no LDS operands, K split over the waves of a workgroup) through the three entry points that
    can take it -- istnet_pw_forward, istnet_pw_forward_acc (accumulators start from a tensor), istnet_pw_forward_ld
    (weights are a column slice of a wider matrix, aligned or not) -- against float64, and against pw_fwd_kernel."""
    lib = _native.lib()
    assert lib.istnet_pw_forward_cfg(b, cin, cout, p) == 1
    g = torch.Generator().manual_seed(b + cin + cout + p)
    x = torch.randn(b, cin, p, generator=g).to(DEV)
    pad = {"ld": 64, "ld_unaligned": 3, "acc": 128}.get(mode, 0)
    wfull = (torch.randn(cout, cin + pad, generator=g) / cin ** 0.5).to(DEV)
    off = pad if mode in ("ld", "acc") else 0           # the slice starts after `pad` columns (or at 0, unaligned row stride)
    w = wfull[:, off:off + cin]
    bn_in = _bn_block(cin, g)
    sc, sh = (bn_in[0].data_ptr(), bn_in[1].data_ptr()) if has_bn else (None, None)
    cinit = torch.randn(b, cout, p, generator=g).to(DEV) if mode == "acc" else None
    d = torch.float64
    act = torch.relu(x * bn_in[0].view(1, -1, 1) + bn_in[1].view(1, -1, 1)).to(d) if has_bn else x.to(d)
    want = torch.matmul(w.to(d), act) + (cinit.to(d) if cinit is not None else 0)
    wptr, ldw = wfull.data_ptr() + 4 * off, cin + pad
    outs = []
    for enable in (1, 0):
        assert lib.istnet_pw_set_tuning(15, enable) == 0
        try:
            nt = lib.istnet_pw_forward_tiles(b, cin, cout, p) if mode == "plain" else lib.istnet_pw_forward_ld_tiles(b, cin, cout, p)
            y = torch.full((b, cout, p), float("nan"), device=DEV)
            part = torch.full((2, cout, nt), float("nan"), device=DEV)
            if mode == "plain":
                rc = lib.istnet_pw_forward(b, cin, cout, p, x.data_ptr(), wptr, sc, sh, y.data_ptr(), part[0].data_ptr(),
                                           part[1].data_ptr(), _st())
            elif mode == "acc":
                rc = lib.istnet_pw_forward_acc(b, cin, cout, p, x.data_ptr(), wptr, ldw, cinit.data_ptr(), y.data_ptr(),
                                               part[0].data_ptr(), part[1].data_ptr(), _st())
            else:
                rc = lib.istnet_pw_forward_ld(b, cin, cout, p, x.data_ptr(), wptr, ldw, sc, sh, y.data_ptr(),
                                              part[0].data_ptr(), part[1].data_ptr(), _st())
            assert rc == 0
            outs.append((y, part))
        finally:
            lib.istnet_pw_set_tuning(15, 1)
    for y, part in outs:
        torch.testing.assert_close(y.to(d), want, rtol=1e-5, atol=2e-5)
        torch.testing.assert_close(part[0].to(d).sum(-1), want.sum(dim=(0, 2)), rtol=1e-5, atol=1e-3)
        torch.testing.assert_close(part[1].to(d).sum(-1), want.square().sum(dim=(0, 2)), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("b,cin_total,ci_off,rows,cout,p,stats", [(4, 512, 0, 512, 512, 128, True), (2, 320, 256, 64, 256, 256, False),
                                                                  (3, 131, 3, 128, 192, 384, False), (2, 96, 0, 40, 64, 128, True),
                                                                  (32, 768, 0, 512, 512, 64, False), (8, 96, 0, 40, 64, 32, True)])
def test_dgrad_split_k_kernel_for_small_launches(b, cin_total, ci_off, rows, cout, p, stats):
    """pw_dgrad_sk_kernel (dense gradient source, small launches) against float64: dA for a column slice of the weight
    matrix, partial row tiles (rows = 40), the BatchNorm-backward statistics partials of the layer below; pw_dgrad_kernel
    (tuning key 17 = 0) agrees."""
    lib = _native.lib()
    assert lib.istnet_pw_dgrad_sk(2, 128, 128, 1024) == 0          # short K keeps the LDS-tiled kernel by default
    assert lib.istnet_pw_set_tuning(18, 64) == 0
    assert lib.istnet_pw_dgrad_sk(b, rows, cout, p) == 1
    g = torch.Generator().manual_seed(b + cin_total + cout + p)
    w = (torch.randn(cout, cin_total, generator=g) / cout ** 0.5).to(DEV)
    y = torch.randn(b, cout, p, generator=g).to(DEV)
    dA = torch.randn(b, cout, p, generator=g).to(DEV)
    bn = _bn_block(cout, g)
    bwdc = torch.stack([torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.01,
                        torch.randn(cout, generator=g) * 0.01]).contiguous().to(DEV)
    y_in = torch.randn(b, rows, p, generator=g).to(DEV)
    bn_in = _bn_block(rows, g)
    d = torch.float64
    mask = (y * bn[0].view(1, -1, 1) + bn[1].view(1, -1, 1)) > 0
    dy = bwdc[0].to(d).view(1, -1, 1) * (dA.to(d) * mask) + bwdc[1].to(d).view(1, -1, 1) + bwdc[2].to(d).view(1, -1, 1) * y.to(d)
    want = torch.matmul(w[:, ci_off:ci_off + rows].to(d).t(), dy)
    gq = want * ((y_in * bn_in[0].view(1, -1, 1) + bn_in[1].view(1, -1, 1)) > 0)
    outs = []
    for enable in (1, 0):
        assert lib.istnet_pw_set_tuning(17, enable) == 0
        try:
            nt = lib.istnet_pw_dgrad_tiles(b, rows, cout, p, 1)
            dx = torch.full((b, rows, p), float("nan"), device=DEV)
            part = torch.full((2, rows, nt), float("nan"), device=DEV)
            pg, pgy = (part[0].data_ptr(), part[1].data_ptr()) if stats else (None, None)
            assert lib.istnet_pw_dgrad(b, cin_total, ci_off, rows, cout, p, 0, w.data_ptr(), y.data_ptr(), dA.data_ptr(), None, 0,
                                       None, bn.data_ptr(), bwdc.data_ptr(), dx.data_ptr(), y_in.data_ptr() if stats else None,
                                       bn_in.data_ptr() if stats else None, pg, pgy, _st()) == 0
            outs.append((dx, part))
        finally:
            lib.istnet_pw_set_tuning(17, 1)
    for dx, part in outs:
        torch.testing.assert_close(dx.to(d), want, rtol=1e-5, atol=2e-5)
        if stats:
            torch.testing.assert_close(part[0].to(d).sum(-1), gq.sum(dim=(0, 2)), rtol=1e-4, atol=2e-3)
            torch.testing.assert_close(part[1].to(d).sum(-1), (gq * y_in.to(d)).sum(dim=(0, 2)), rtol=1e-4, atol=2e-3)
    assert lib.istnet_pw_set_tuning(18, 0) == 0


@pytest.mark.parametrize("b,p,pooled", [(3, 512, True), (2, 1024, False), (32, 2048, True)])
def test_dgrad_loader_mfma_roles_for_256_output_channels(b, p, pooled):
    """istnet_pw_dgrad at cout = 256, cin = 128 with statistics: the dgrad-only mode of pw_bwd_mid_kernel (loader / MFMA
    wave roles) against float64 and against pw_dgrad_kernel (tuning key 19 = 0), dense and pooled gradient source."""
    lib = _native.lib()
    cin, cout, s = 128, 256, 16
    assert lib.istnet_pw_set_tuning(17, 0) == 0          # (keep the small dense cases off the split-K kernel)
    try:
        assert lib.istnet_pw_dgrad_rs(b, cin, cout, p, 0 if pooled else 1) == 1
        g = torch.Generator().manual_seed(b + p)
        w = (torch.randn(cout, cin, generator=g) / cout ** 0.5).to(DEV)
        y = torch.randn(b, cout, p, generator=g).to(DEV)
        bn, bn_in = _bn_block(cout, g), _bn_block(cin, g)
        bwdc = torch.stack([torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.01,
                            torch.randn(cout, generator=g) * 0.01]).contiguous().to(DEV)
        y_in = torch.randn(b, cin, p, generator=g).to(DEV)
        if pooled:
            gr = p // s
            dpool = torch.randn(b, cout, gr, generator=g).to(DEV)
            arg = torch.randint(0, s, (b, cout, gr), generator=g, dtype=torch.uint8).to(DEV)
            src = (None, dpool.data_ptr(), 0, arg.data_ptr())
            gdense = torch.zeros(b, cout, gr, s, device=DEV).scatter_(3, arg.long().unsqueeze(-1), dpool.unsqueeze(-1)).reshape(b, cout, p)
        else:
            gdense = torch.randn(b, cout, p, generator=g).to(DEV)
            src = (gdense.data_ptr(), None, 0, None)
        d = torch.float64
        mask = (y * bn[0].view(1, -1, 1) + bn[1].view(1, -1, 1)) > 0
        dy = bwdc[0].to(d).view(1, -1, 1) * (gdense.to(d) * mask) + bwdc[1].to(d).view(1, -1, 1) + bwdc[2].to(d).view(1, -1, 1) * y.to(d)
        want = torch.matmul(w.to(d).t(), dy)
        gq = want * ((y_in * bn_in[0].view(1, -1, 1) + bn_in[1].view(1, -1, 1)) > 0)
        outs = []
        for enable in (1, 0):
            assert lib.istnet_pw_set_tuning(19, enable) == 0
            try:
                nt = lib.istnet_pw_dgrad_tiles(b, cin, cout, p, 0 if pooled else 1)
                dx = torch.full((b, cin, p), float("nan"), device=DEV)
                part = torch.full((2, cin, nt), float("nan"), device=DEV)
                assert lib.istnet_pw_dgrad(b, cin, 0, cin, cout, p, s if pooled else 0, w.data_ptr(), y.data_ptr(), *src,
                                           bn.data_ptr(), bwdc.data_ptr(), dx.data_ptr(), y_in.data_ptr(), bn_in.data_ptr(),
                                           part[0].data_ptr(), part[1].data_ptr(), _st()) == 0
                outs.append((dx, part))
            finally:
                lib.istnet_pw_set_tuning(19, 1)
        for dx, part in outs:
            torch.testing.assert_close(dx.to(d), want, rtol=1e-5, atol=2e-5)
            torch.testing.assert_close(part[0].to(d).sum(-1), gq.sum(dim=(0, 2)), rtol=1e-4, atol=2e-3)
            torch.testing.assert_close(part[1].to(d).sum(-1), (gq * y_in.to(d)).sum(dim=(0, 2)), rtol=1e-4, atol=2e-3)
    finally:
        lib.istnet_pw_set_tuning(17, 1)


@pytest.mark.parametrize("b,c1,cout,n,m", [(4, 256, 512, 128, 64), (2, 64, 72, 256, 100), (3, 128, 256, 384, 192), (4, 64, 96, 64, 32)])
def test_forward_acc_with_the_interpolation_in_the_epilogue(b, c1, cout, n, m):
    """istnet_pw_forward_acc_interp: y = three_interpolate(zk, idx, weight) + w . x in one launch, against float64, with the
    statistics partials; same result as interpolating first and calling istnet_pw_forward_acc."""
    from istnet_amd.pointnet2 import _ext
    lib = _native.lib()
    assert lib.istnet_pw_forward_cfg(b, c1, cout, n) == 1
    g = torch.Generator().manual_seed(b + c1 + cout + n)
    x = torch.randn(b, c1, n, generator=g).to(DEV)
    ldw = c1 + 32
    wfull = (torch.randn(cout, ldw, generator=g) / c1 ** 0.5).to(DEV)
    zk = torch.randn(b, cout, m, generator=g).to(DEV)
    idx = torch.randint(0, m, (b, n, 3), generator=g, dtype=torch.int32).to(DEV)
    wt = torch.rand(b, n, 3, generator=g)
    wt = (wt / wt.sum(-1, keepdim=True)).to(DEV)
    nt = lib.istnet_pw_forward_ld_tiles(b, c1, cout, n)
    y = torch.full((b, cout, n), float("nan"), device=DEV)
    part = torch.full((2, cout, nt), float("nan"), device=DEV)
    assert lib.istnet_pw_forward_acc_interp(b, c1, cout, n, x.data_ptr(), wfull.data_ptr() + 4 * 32, ldw, zk.data_ptr(), m,
                                            idx.data_ptr(), wt.data_ptr(), y.data_ptr(), part[0].data_ptr(),
                                            part[1].data_ptr(), _st()) == 0
    d = torch.float64
    gathered = torch.gather(zk.to(d).unsqueeze(2).expand(-1, -1, n, -1), 3,
                            idx.long().unsqueeze(1).expand(-1, cout, -1, -1))          # (b, cout, n, 3)
    want = (gathered * wt.to(d).unsqueeze(1)).sum(-1) + torch.matmul(wfull[:, 32:].to(d), x.to(d))
    torch.testing.assert_close(y.to(d), want, rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(part[0].to(d).sum(-1), want.sum(dim=(0, 2)), rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(part[1].to(d).sum(-1), want.square().sum(dim=(0, 2)), rtol=1e-5, atol=1e-3)
    t_int = _ext.three_interpolate(zk, idx, wt)
    y2 = torch.empty_like(y)
    assert lib.istnet_pw_forward_acc(b, c1, cout, n, x.data_ptr(), wfull.data_ptr() + 4 * 32, ldw, t_int.data_ptr(),
                                     y2.data_ptr(), None, None, _st()) == 0
    torch.testing.assert_close(y, y2, rtol=1e-6, atol=2e-6)
    assert lib.istnet_pw_forward_acc_interp(32, c1, cout, 32768, x.data_ptr(), wfull.data_ptr(), ldw, zk.data_ptr(), m,
                                            idx.data_ptr(), wt.data_ptr(), y.data_ptr(), None, None, _st()) != 0   # too large: refused


def test_forward_acc_channel_stats_and_dy():
    lib = _native.lib()
    b, cin, cout, p = 2, 24, 40, 256
    g = torch.Generator().manual_seed(9)
    x = torch.randn(b, cin, p, generator=g).to(DEV)
    w = (torch.randn(cout, cin + 8, generator=g) * 0.2).to(DEV)          # the layer uses columns 8.. of a wider matrix
    t = torch.randn(b, cout, p, generator=g).to(DEV)
    y = torch.empty(b, cout, p, device=DEV)
    nt = lib.istnet_pw_stat_tiles(b, cout, p)
    part = torch.empty(2, cout, nt, device=DEV)
    assert lib.istnet_pw_forward_acc(b, cin, cout, p, x.data_ptr(), w.data_ptr() + 32, cin + 8, t.data_ptr(), y.data_ptr(),
                                     part[0].data_ptr(), part[1].data_ptr(), _st()) == 0
    want = t + torch.matmul(w[:, 8:], x)
    torch.testing.assert_close(y, want, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(part[0].sum(-1), want.sum(dim=(0, 2)), rtol=1e-4, atol=1e-3)
    nt2 = lib.istnet_pw_bwd_stat_tiles(b, p)
    part2 = torch.empty(2, cout, nt2, device=DEV)
    assert lib.istnet_pw_channel_stats(b, cout, p, y.data_ptr(), part2[0].data_ptr(), part2[1].data_ptr(), _st()) == 0
    torch.testing.assert_close(part2[1].sum(-1), want.square().sum(dim=(0, 2)), rtol=1e-4, atol=1e-3)
    bn = _bn_block(cout, g)
    bwdc = torch.stack([torch.rand(cout, generator=g), torch.randn(cout, generator=g), torch.randn(cout, generator=g)]).contiguous().to(DEV)
    d = torch.randn(b, cout, p, generator=g).to(DEV)
    out = torch.empty_like(y)
    assert lib.istnet_pw_dy(b, cout, p, y.data_ptr(), d.data_ptr(), bn.data_ptr(), bwdc.data_ptr(), out.data_ptr(), _st()) == 0
    mask = (y * bn[0].view(1, -1, 1) + bn[1].view(1, -1, 1)) > 0
    torch.testing.assert_close(out, bwdc[0].view(1, -1, 1) * (d * mask) + bwdc[1].view(1, -1, 1) + bwdc[2].view(1, -1, 1) * y,
                               rtol=1e-5, atol=1e-5)


def test_scatter_dwx_partials():
    """istnet_pw_scatter_dy: the scattered dY0 and, in both modes, the xyz-weight partials."""
    lib = _native.lib()
    b, n, npoint, s, cout = 2, 128, 32, 16, 12
    g = torch.Generator().manual_seed(5)
    xyz = torch.rand(b, n, 3, generator=g).to(DEV)
    new_xyz = xyz[:, :npoint].contiguous()
    idx = torch.randint(0, n, (b, npoint, s), generator=g, dtype=torch.int32).to(DEV)
    p = npoint * s
    y = torch.randn(b, cout, p, generator=g).to(DEV)
    d = torch.randn(b, cout, p, generator=g).to(DEV)
    bn = _bn_block(cout, g)
    bwdc = torch.stack([torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1,
                        torch.randn(cout, generator=g) * 0.1]).contiguous().to(DEV)
    mask = (y * bn[0].view(1, -1, 1) + bn[1].view(1, -1, 1)) > 0
    dy = bwdc[0].view(1, -1, 1) * (d * mask) + bwdc[1].view(1, -1, 1) + bwdc[2].view(1, -1, 1) * y
    flat = idx.long().reshape(b, p)
    xrel = torch.gather(xyz, 1, flat.unsqueeze(-1).expand(-1, -1, 3)) - new_xyz.repeat_interleave(s, dim=1)
    want_g = torch.zeros(b, cout, n, device=DEV).scatter_add_(2, flat.unsqueeze(1).expand(-1, cout, -1), dy)
    want_dwx = torch.einsum("bcp,bpk->ck", dy, xrel)
    out = torch.empty(b, cout, n, device=DEV)
    dwx = torch.empty(b, cout, 3, device=DEV)
    assert lib.istnet_pw_scatter_dy(b, cout, n, p, 0, y.data_ptr(), d.data_ptr(), None, 0, None, bn.data_ptr(),
                                    bwdc.data_ptr(), idx.data_ptr(), out.data_ptr(), 0, xyz.data_ptr(),
                                    new_xyz.data_ptr(), s, dwx.data_ptr(), _st()) == 0
    torch.testing.assert_close(out, want_g, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dwx.sum(0), want_dwx, rtol=1e-4, atol=1e-4)
    chunks = lib.istnet_pw_dwx_chunks(b, cout, p)
    dwx2 = torch.empty(b * chunks, cout, 3, device=DEV)
    assert lib.istnet_pw_scatter_dy(b, cout, n, p, 0, y.data_ptr(), d.data_ptr(), None, 0, None, bn.data_ptr(),
                                    bwdc.data_ptr(), idx.data_ptr(), None, 0, xyz.data_ptr(), new_xyz.data_ptr(), s,
                                    dwx2.data_ptr(), _st()) == 0
    torch.testing.assert_close(dwx2.sum(0), want_dwx, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("n,npoint,s,cout", [(128, 32, 16, 12), (512, 256, 32, 32), (300, 64, 8, 5), (1024, 512, 16, 64)])
def test_scatter_csr_matches_dense_scatter_and_is_deterministic(n, npoint, s, cout):
    """istnet_pw_scatter_dy_csr (inverse lists, no atomics) == scatter_add of the dense dY0, same dwx, and two runs
    are bit-identical."""
    from istnet_amd.pointnet2 import _ext
    lib = _native.lib()
    b = 3
    g = torch.Generator().manual_seed(n + s)
    xyz = torch.rand(b, n, 3, generator=g).to(DEV)
    new_xyz = xyz[:, :npoint].contiguous()
    idx = torch.sort(torch.randint(0, n, (b, npoint, s), generator=g), dim=2).values.int()
    idx[:, :, s // 2:] = idx[:, :, s // 2 - 1:s // 2]          # padded rows: the tail repeats one index
    idx = idx.to(DEV).contiguous()
    p = npoint * s
    y = torch.randn(b, cout, p, generator=g).to(DEV)
    d = torch.randn(b, cout, p, generator=g).to(DEV)
    bn = _bn_block(cout, g)
    bwdc = torch.stack([torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1,
                        torch.randn(cout, generator=g) * 0.1]).contiguous().to(DEV)
    off, ent = _ext.ball_csr(idx, n)
    flat = idx.long().reshape(b, p)
    # lists: ascending slots, covering every slot once, consistent with idx
    assert int(off[:, -1].min()) == p and torch.equal(torch.sort(ent, dim=1).values, torch.arange(p, device=DEV).expand(b, p).int())
    assert torch.equal(torch.gather(flat, 1, ent.long()), torch.repeat_interleave(torch.arange(n, device=DEV), 1).expand(b, n)
                       .repeat_interleave(1, dim=1).gather(1, torch.searchsorted(off[:, 1:].contiguous().long(), torch.arange(p, device=DEV).expand(b, p).contiguous(), right=True)))
    mask = (y * bn[0].view(1, -1, 1) + bn[1].view(1, -1, 1)) > 0
    dy = bwdc[0].view(1, -1, 1) * (d * mask) + bwdc[1].view(1, -1, 1) + bwdc[2].view(1, -1, 1) * y
    xrel = torch.gather(xyz, 1, flat.unsqueeze(-1).expand(-1, -1, 3)) - new_xyz.repeat_interleave(s, dim=1)
    want_g = torch.zeros(b, cout, n, device=DEV).scatter_add_(2, flat.unsqueeze(1).expand(-1, cout, -1), dy)
    want_dwx = torch.einsum("bcp,bpk->ck", dy, xrel)
    chunks = lib.istnet_pw_scatter_csr_chunks(n)
    first = None
    try:
        for threads in (0, 256, 512, 1024):              # key 21: workgroup size (0 = by the cloud's size)
            assert lib.istnet_pw_set_tuning(21, threads) == 0
            outs = []
            for _ in range(2):
                out = torch.empty(b, cout, n, device=DEV)
                dwx = torch.empty(b * chunks, cout, 3, device=DEV)
                assert lib.istnet_pw_scatter_dy_csr(b, cout, n, p, y.data_ptr(), d.data_ptr(), bn.data_ptr(), bwdc.data_ptr(),
                                                    off.data_ptr(), ent.data_ptr(), out.data_ptr(), 0, xyz.data_ptr(),
                                                    new_xyz.data_ptr(), s, dwx.data_ptr(), _st()) == 0
                outs.append((out, dwx))
            torch.testing.assert_close(outs[0][0], want_g, rtol=1e-4, atol=1e-4)
            torch.testing.assert_close(outs[0][1].sum(0), want_dwx, rtol=1e-4, atol=2e-4)
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
            if first is None:
                first = outs[0][0]
            assert torch.equal(outs[0][0], first)        # a point's list is summed in the same order whatever the workgroup size
    finally:
        lib.istnet_pw_set_tuning(21, 0)


@pytest.mark.parametrize("b,c,n,m", [(2, 128, 1024, 512), (3, 20, 256, 100), (2, 8, 64, 16), (1, 260, 128, 64)])
def test_interp_grad_with_dy_formed_per_element(b, c, n, m):
    """istnet_interp_grad_csr_dy: the gradient of three_interpolate's input with dY = ca g [y scale + shift > 0] + cb + cc y
    formed per gathered element -- from global memory (key 22 = 0) and from LDS-staged rows (default) -- against
    three_interpolate's scatter of the materialised dY; the two variants bit-identical; ragged sizes."""
    from istnet_amd.pointnet2 import _ext
    lib = _native.lib()
    g = torch.Generator().manual_seed(c + n)
    unknown, known = torch.rand(b, n, 3, generator=g).to(DEV), torch.rand(b, m, 3, generator=g).to(DEV)
    idx, weight = _ext.three_nn_weights(unknown, known)
    off, ent = _ext.interp_csr(idx, m)
    y = torch.randn(b, c, n, generator=g).to(DEV)
    d = torch.randn(b, c, n, generator=g).to(DEV)
    bn = _bn_block(c, g)
    bwdc = torch.stack([torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1,
                        torch.randn(c, generator=g) * 0.1]).contiguous().to(DEV)
    mask = (y * bn[0].view(1, -1, 1) + bn[1].view(1, -1, 1)) > 0
    dy = (bwdc[0].view(1, -1, 1) * (d * mask) + bwdc[1].view(1, -1, 1) + bwdc[2].view(1, -1, 1) * y).double()
    want = torch.zeros(b, c, m, dtype=torch.float64, device=DEV)
    for k in range(3):
        want.scatter_add_(2, idx[:, :, k].long().unsqueeze(1).expand(-1, c, -1), dy * weight[:, :, k].double().unsqueeze(1))
    outs = []
    try:
        for flag in (0, 1):
            assert lib.istnet_pw_set_tuning(22, flag) == 0
            out = torch.empty(b, c, m, device=DEV)
            assert lib.istnet_interp_grad_csr_dy(b, c, n, m, y.data_ptr(), d.data_ptr(), bn.data_ptr(), bwdc.data_ptr(),
                                                 weight.data_ptr(), off.data_ptr(), ent.data_ptr(), out.data_ptr(), _st()) == 0
            outs.append(out)
    finally:
        lib.istnet_pw_set_tuning(22, 1)
    torch.testing.assert_close(outs[1].double(), want, rtol=1e-5, atol=1e-5)
    assert torch.equal(outs[0], outs[1])


def test_inverse_lists_on_degenerate_indices():
    """Every slot pointing at the same source point (one list of length E), and a permutation (all lists length 1):
    the stable build returns ascending entries in bounded time."""
    from istnet_amd.pointnet2 import _ext
    b, npoint, s, n = 2, 256, 32, 512
    same = torch.full((b, npoint, s), 7, dtype=torch.int32, device=DEV)
    off, ent = _ext.ball_csr(same, n)
    e = npoint * s
    assert torch.equal(ent, torch.arange(e, device=DEV, dtype=torch.int32).expand(b, e))
    assert int(off[0, 7]) == 0 and int(off[0, 8]) == e and int(off[0, -1]) == e
    perm = torch.stack([torch.randperm(n, generator=torch.Generator().manual_seed(k)) for k in range(b)]).int().to(DEV)
    off, ent = _ext.ball_csr(perm.view(b, n // 16, 16).contiguous(), n)
    assert torch.equal(off, torch.arange(n + 1, device=DEV, dtype=torch.int32).expand(b, n + 1))
    assert torch.equal(torch.gather(perm.long(), 1, ent.long()), torch.arange(n, device=DEV).expand(b, n))


def _lists_reference(idx2d, m):
    """Inverse lists by a stable sort on the host: offsets (B, m+1), entries (B, E) ascending inside each list."""
    b, e = idx2d.shape
    keys = idx2d.long().cpu()
    ent = torch.argsort(keys, dim=1, stable=True).int()
    off = torch.zeros(b, m + 1, dtype=torch.int32)
    for bi in range(b):
        off[bi, 1:] = torch.cumsum(torch.bincount(keys[bi], minlength=m), 0).int()
    return off, ent


@pytest.mark.parametrize("b,shapes", [
    (3, [((256, 16), 512), ((256, 32), 512), ((128, 16), 256), ((64, 32), 128), ((1024, 3), 512), ((128, 3), 64)]),
    (32, [((256, 32), 512), ((512, 3), 256)]),
    (2, [((5, 3), 1), ((1, 1), 70), ((300, 7), 130), ((3000, 5), 4000)]),
])
def test_inverse_lists_key_range_kernel_vs_stable_sort(b, shapes):
    """istnet_pn2_csr_build_multi (a workgroup per 64 source points, several index tensors per launch) against a stable
    argsort on the host and against the one-workgroup-per-cloud kernels: identical offsets and entries, on ball-query-like
    rows (ascending with a padded tail), on three_nn-like taps, on a key count that is not a multiple of 64, on more
    keys than slots, and on slot counts that are not multiples of 256."""
    from istnet_amd.pointnet2 import _ext
    lib = _native.lib()
    g = torch.Generator().manual_seed(b)
    problems = []
    for (rows, s), m in shapes:
        idx = torch.sort(torch.randint(0, m, (b, rows, s), generator=g), dim=2).values
        if s >= 8:
            idx[:, ::3, s // 2:] = idx[:, ::3, :1]          # padded rows repeat the first hit
            idx[:, :, -1] = idx[:, :1, 0]                    # one source point referenced by every row (a long list)
        problems.append((idx.int().to(DEV).contiguous(), m))
    got = _ext.csr_multi(problems)
    torch.cuda.synchronize()
    assert lib.istnet_pn2_set_tuning(2, 1) == 0
    try:
        legacy = [_ext.csr_multi([p])[0] for p in problems]
        torch.cuda.synchronize()
    finally:
        assert lib.istnet_pn2_set_tuning(2, 0) == 0
    for (idx, m), (off, ent), old in zip(problems, got, legacy):
        want_off, want_ent = _lists_reference(idx.reshape(b, -1), m)
        assert torch.equal(off.cpu(), want_off) and torch.equal(ent.cpu(), want_ent)
        if old is not None:
            assert torch.equal(old[0], off) and torch.equal(old[1], ent)


def test_inverse_lists_large_slot_count_takes_the_per_cloud_kernel():
    """More slots than the range kernel's LDS queue holds (N=2048 level 1: 1024 x 32 slots): the per-cloud kernels
    build the same lists."""
    from istnet_amd.pointnet2 import _ext
    g = torch.Generator().manual_seed(9)
    idx = torch.randint(0, 2048, (2, 1024, 32), generator=g).int().to(DEV)
    off, ent = _ext.ball_csr(idx, 2048)
    want_off, want_ent = _lists_reference(idx.reshape(2, -1), 2048)
    assert torch.equal(off.cpu(), want_off) and torch.equal(ent.cpu(), want_ent)


def test_row_kernels_beyond_the_grid_y_limit():
    """B * C = 81 920 rows (> 65 535, the limit of grid.y): the BN + ReLU (+ max-pool) tail and the dY kernel loop over
    rows instead of failing the launch."""
    lib = _native.lib()
    b, c, g, s = 160, 512, 4, 4
    gen = torch.Generator().manual_seed(9)
    y = torch.randn(b, c, g * s, generator=gen).to(DEV)
    bn = _bn_block(c, gen)
    out = torch.empty(b, c, g, device=DEV)
    arg = torch.empty(b, c, g, dtype=torch.uint8, device=DEV)
    ymax = torch.empty(b, c, g, device=DEV)
    assert lib.istnet_bn_relu_pool(b, c, g, s, y.data_ptr(), bn.data_ptr(), out.data_ptr(), 0, arg.data_ptr(), ymax.data_ptr(),
                                   _st()) == 0
    act = torch.relu(y * bn[0].view(1, -1, 1) + bn[1].view(1, -1, 1)).view(b, c, g, s)
    torch.testing.assert_close(out, act.amax(dim=3), rtol=0, atol=0)
    dense = torch.empty(b, c, g * s, device=DEV)
    assert lib.istnet_bn_relu_pool(b, c, g * s, 1, y.data_ptr(), bn.data_ptr(), dense.data_ptr(), 0, None, None, _st()) == 0
    torch.testing.assert_close(dense, act.view(b, c, g * s), rtol=0, atol=0)
    d = torch.randn(b, c, g * s, generator=gen).to(DEV)
    bwdc = torch.stack([torch.rand(c, generator=gen) + 0.5, torch.randn(c, generator=gen) * 0.01,
                        torch.randn(c, generator=gen) * 0.01]).contiguous().to(DEV)
    dy = torch.empty_like(y)
    assert lib.istnet_pw_dy(b, c, g * s, y.data_ptr(), d.data_ptr(), bn.data_ptr(), bwdc.data_ptr(), dy.data_ptr(), _st()) == 0
    want = bwdc[0].view(1, -1, 1) * (d * (act.view(b, c, -1) > 0)) + bwdc[1].view(1, -1, 1) + bwdc[2].view(1, -1, 1) * y
    torch.testing.assert_close(dy, want, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("b,c,m,n", [(2, 20, 64, 128), (3, 128, 96, 300), (1, 7, 32, 256)])
def test_interp_stats_is_three_interpolate_plus_channel_sums(b, c, m, n):
    """istnet_pw_interp_stats: the interpolated tensor is bit-identical to the stand-alone three_interpolate (reference
    interpolate_gpu.cu:77-106) and the partials sum to the per-channel sums of that tensor (ragged n and c included)."""
    from istnet_amd.pointnet2 import _ext
    lib = _native.lib()
    g = torch.Generator().manual_seed(17 * b + n)
    pts = torch.randn(b, c, m, generator=g).to(DEV)
    idx = torch.randint(0, m, (b, n, 3), generator=g, dtype=torch.int32).to(DEV)
    wt = torch.rand(b, n, 3, generator=g)
    wt = (wt / wt.sum(-1, keepdim=True)).to(DEV)
    out = torch.empty(b, c, n, device=DEV)
    nt = lib.istnet_pw_interp_stats_tiles(b, n)
    part = torch.full((2, c, nt), float("nan"), device=DEV)
    assert lib.istnet_pw_interp_stats(b, c, m, n, pts.data_ptr(), idx.data_ptr(), wt.data_ptr(), out.data_ptr(),
                                      part[0].data_ptr(), part[1].data_ptr(), _st()) == 0
    want = _ext.three_interpolate(pts, idx, wt)
    assert torch.equal(out, want)
    w64 = want.double()
    torch.testing.assert_close(part[0].double().sum(-1), w64.sum(dim=(0, 2)), rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(part[1].double().sum(-1), w64.square().sum(dim=(0, 2)), rtol=1e-5, atol=1e-4)
    assert lib.istnet_pw_interp_stats(b, c, m, n, None, idx.data_ptr(), wt.data_ptr(), out.data_ptr(),
                                      part[0].data_ptr(), part[1].data_ptr(), _st()) != 0
