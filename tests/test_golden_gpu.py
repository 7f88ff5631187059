"""GPU parity against the golden vectors produced by the reference Python (tests/golden), plus
size-independent properties at the BASELINE.json full size (B=32, N=1024).

Bar: index tensors bit-exact; features / poses within 1e-4 (fp32), as north_star states."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")
CAM = [[0.01, 0.02], [0.02, 0.04], [0.04, 0.08], [0.08, 0.16]]
TOL = dict(rtol=1e-4, atol=1e-4)


def test_config1_single_sa_layer():
    """BASELINE config 1: ball_query r=0.2 nsample=32 on B=4 N=1024."""
    from istnet_amd.pointnet2 import pointnet2_utils as pu
    z = np.load(os.path.join(GOLD, "config1_sa_grouping.npz"))
    xyz = torch.from_numpy(z["xyz"]).to(DEV)
    fps = pu.furthest_point_sample(xyz, 512)
    assert np.array_equal(fps.cpu().numpy(), z["fps_idx"].astype(np.int32))
    new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), fps).transpose(1, 2).contiguous()
    assert np.array_equal(new_xyz.cpu().numpy(), z["new_xyz"])
    idx = pu.ball_query(0.2, 32, xyz, new_xyz)
    assert np.array_equal(idx.cpu().numpy(), z["ball_idx"].astype(np.int32))
    grouped = pu.QueryAndGroup(0.2, 32)(xyz, new_xyz, None)
    assert np.array_equal(grouped.cpu().numpy()[:, :, ::16], z["grouped_xyz"])


def test_sa_fp_layer():
    from istnet_amd.pointnet2.pointnet2_modules import PointnetFPModule, PointnetSAModuleMSG
    z = np.load(os.path.join(GOLD, "sa_fp_layer.npz"))
    torch.manual_seed(20)
    sa = PointnetSAModuleMSG(npoint=64, radii=[0.15, 0.3], nsamples=[8, 16], mlps=[[16, 16, 32], [16, 16, 32]]).to(DEV)
    fp = PointnetFPModule(mlp=[64 + 16, 32, 32]).to(DEV)
    xyz = torch.from_numpy(z["xyz"]).to(DEV)
    feat = torch.from_numpy(z["feat"]).to(DEV).requires_grad_(True)
    new_xyz, new_feat = sa(xyz, feat)
    out = fp(xyz, new_xyz, feat, new_feat)
    out.square().mean().backward()
    assert np.array_equal(new_xyz.detach().cpu().numpy(), z["new_xyz"])
    np.testing.assert_allclose(new_feat.detach().cpu().numpy(), z["sa_out"], **TOL)
    np.testing.assert_allclose(out.detach().cpu().numpy(), z["fp_out"], **TOL)
    np.testing.assert_allclose(feat.grad.cpu().numpy(), z["grad_feat"], **TOL)
    for name, p in list(sa.named_parameters()) + list(fp.named_parameters()):
        # measured (round 5): max |difference| 1.1e-8 on gradients of magnitude 1e-4 .. 1e-3 -- two fp32 evaluations; the
        # float64 comparison below is the bar that counts
        np.testing.assert_allclose(p.grad.cpu().numpy(), z["gradp_" + name], rtol=2e-4, atol=1e-7, err_msg=name)
    # the bar above compares two float32 evaluations (the golden is the reference's fp32 run).  Against a float64 evaluation
    # of the same modules with the same index decisions every parameter gradient holds 1e-4 of its own norm, except where a
    # max-pool candidate within round-off of the maximum re-routes a gradient (at most one layer may carry such a flip here)
    import copy
    from istnet_amd.pointnet2 import pointnet2_utils
    sa64, fp64 = copy.deepcopy(sa).double(), copy.deepcopy(fp).double()
    for m_ in list(sa64.modules()) + list(fp64.modules()):
        if isinstance(m_, torch.nn.modules.batchnorm._BatchNorm):
            m_.reset_running_stats()
    saved = pointnet2_utils._ext
    try:
        pointnet2_utils._ext = _F64Ext(saved)
        feat64 = torch.from_numpy(z["feat"]).to(DEV).double().requires_grad_(True)
        xyz64 = xyz.double()
        nx64, nf64 = sa64(xyz64, feat64)
        fp64(xyz64, nx64, feat64, nf64).square().mean().backward()
    finally:
        pointnet2_utils._ext = saved
    rel = lambda a, b_: float((a.double() - b_).norm() / b_.norm().clamp_min(1e-30))
    assert rel(feat.grad, feat64.grad) < 1e-4
    errs = {n_: rel(p.grad, q.grad) for (n_, p), q in zip(list(sa.named_parameters()) + list(fp.named_parameters()),
                                                          list(sa64.parameters()) + list(fp64.parameters()))}
    # measured (round 5): the three largest are 1.4e-6, 1.6e-6, 1.7e-5 -- no max-pool flip on this input, so the plain
    # 1e-4 bar holds for every parameter (rounds 2-4 allowed one layer 2e-3 for a flip that this fixture does not have)
    assert max(errs.values()) < 1e-4, errs


def test_encoder_b2(monkeypatch):
    from istnet_amd.modules import PointNet2MSG
    from istnet_amd.pointnet2 import _ext
    z = np.load(os.path.join(GOLD, "encoder_b2.npz"))
    captured, counters = {}, {}

    def tap(name):
        orig = getattr(_ext, name)

        def fn(*a, **k):
            res = orig(*a, **k)
            i = counters.get(name, 0)
            counters[name] = i + 1
            captured[f"{name}_{i}"] = res
            return res
        monkeypatch.setattr(_ext, name, fn)
    # the encoder samples + gathers in one op (chained over the levels: a level whose parent run had no arg-max tie
    # takes the prefix of its input), and takes three_nn together with the interpolation weights
    # ... and asks for the two radii of a level in one launch (ball_query_pair: index tensors 2l and 2l + 1 of the golden)
    # ... and the neighbour searches of the four propagation levels in one launch (three_nn_weights_multi, coarse level first)
    for n in ("furthest_point_sampling_chain", "ball_query_pair", "three_nn_weights_multi"):
        tap(n)
    torch.manual_seed(0)
    enc = PointNet2MSG([list(r) for r in CAM]).to(DEV).train()
    pts = torch.from_numpy(z["pts"]).to(DEV)
    out = enc(pts)
    for lvl in range(4):
        captured[f"ball_query_{2 * lvl}"], captured[f"ball_query_{2 * lvl + 1}"] = captured[f"ball_query_pair_{lvl}"][:2]
        captured[f"three_nn_weights_{lvl}"] = captured["three_nn_weights_multi_0"][lvl]
    for i in range(4):
        assert np.array_equal(captured[f"furthest_point_sampling_chain_{i}"][0].cpu().numpy(),
                              z[f"furthest_point_sampling_{i}"].astype(np.int32)), i
        idx, wgt = captured[f"three_nn_weights_{i}"]
        assert np.array_equal(idx.cpu().numpy(), z[f"three_nn_idx_{i}"].astype(np.int32)), i
        # the weights follow from the golden distances; the squared distances themselves bit for bit from the
        # stand-alone op on the same level coordinates (call i of the encoder is propagation level 3 - i)
        levels = [pts[:, :, :3].contiguous()] + [captured[f"furthest_point_sampling_chain_{k}"][1] for k in range(4)]
        d2, idx_again = _ext.three_nn(levels[3 - i].contiguous(), levels[4 - i].contiguous())
        assert np.array_equal(d2.cpu().numpy(), z[f"three_nn_dist2_{i}"]), i
        assert torch.equal(idx_again, idx)
        inv = 1.0 / (np.sqrt(z[f"three_nn_dist2_{i}"].astype(np.float32)) + np.float32(1e-8))
        np.testing.assert_allclose(wgt.cpu().numpy(), inv / inv.sum(axis=2, keepdims=True), rtol=3e-6, atol=1e-7)
    for i in range(8):
        assert np.array_equal(captured[f"ball_query_{i}"].cpu().numpy(), z[f"ball_query_{i}"].astype(np.int32)), i
    # The reference's own fp32 output is up to 8.7e-5 away from a float64 evaluation of the same model with the
    # same index decisions (make_golden.py prints it), so the 1e-4 bar is applied against that float64 result
    # (out_train_f64); against the reference's fp32 output two valid fp32 evaluations may differ by the sum of
    # their rounding errors.
    # Round 2 (profiles/r02_error_budget_golden_b2.txt): the worst output of the HIP path is 7.9e-5 from the float64
    # result (the fp32 torch composition of the same layers: 1.14e-4), so the plain 1e-4 bar holds here as well.
    got = out.detach().cpu().numpy()[:, :, ::8]
    np.testing.assert_allclose(got, z["out_train_f64"], **TOL)
    np.testing.assert_allclose(got, z["out_train"], rtol=2e-4, atol=2e-4)
    out.square().mean().backward()
    # WIRING CHECK ONLY (every parameter receives a gradient of the right size): the golden is the reference's own fp32
    # backward, and gradients of this loss through 16 train-mode BatchNorm layers are ill-conditioned end to end -- two valid
    # fp32 evaluations (torch-CPU vs torch-GPU composition) already differ by 0.3-1 % on several parameters.  The accuracy
    # bar for gradients is test_encoder_parameter_gradients_vs_float64_autograd below (every parameter gradient of the HIP
    # path against float64 autograd with the same index decisions), not this comparison.
    norms = np.array([float(p.grad.double().norm()) for _, p in enc.named_parameters()])
    np.testing.assert_allclose(norms, z["grad_norms"], rtol=3e-2, atol=1e-7)
    g = dict(enc.named_parameters())
    # measured (round 5) against the reference's fp32 gradients: 1.3e-5 for the last layer (one GEMM away from the loss),
    # 3.4e-3 for the first convolution (the whole network away: the ill-conditioning described above)
    for key, name, bound in (("SA_modules.0.mlps.0.layer0.conv.weight", "grad_first_conv", 1e-2),
                             ("FP_modules.0.mlp.layer1.conv.weight", "grad_last_fp_conv", 1e-4)):
        got, want = g[key].grad.cpu().numpy(), z[name]
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < bound, key
    sd = enc.state_dict()
    np.testing.assert_allclose(sd["SA_modules.3.mlps.1.layer2.normlayer.bn.running_mean"].cpu().numpy(),
                               z["running_mean_sa3"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(sd["SA_modules.3.mlps.1.layer2.normlayer.bn.running_var"].cpu().numpy(),
                               z["running_var_sa3"], rtol=1e-4, atol=1e-7)
    enc.eval()
    with torch.no_grad():
        out_eval = enc(pts)
    np.testing.assert_allclose(out_eval.cpu().numpy()[:, :, ::8], z["out_eval"], **TOL)


@pytest.mark.parametrize("conv", [0, 1, 2])
def test_index_goldens_under_each_fma_convention(conv, monkeypatch):
    """DESIGN.md section 4 / INTEGRATION.md A: the index-deciding squared distances under the source-order convention (0,
    the default) and under the two FMA contractions an nvcc -O3 build of the reference may use (1 = LLVM's contraction of this
    expression tree, the likely one; 2).  tests/golden/index_conventions.npz holds, per convention, what the REFERENCE's
    Python produced over the CPU oracle in that convention (make_golden_conventions.py): config 1's FPS / ball-query
    indices and every index tensor of the encoder on four cube clouds, one of which flips 138 entries between conventions.
    The HIP kernels, switched with istnet_pn2_set_tuning(1, conv), must reproduce each set bit for bit."""
    from istnet_amd import _native
    from istnet_amd.modules import PointNet2MSG
    from istnet_amd.pointnet2 import _ext
    z = np.load(os.path.join(GOLD, "index_conventions.npz"))
    lib = _native.lib()
    assert lib.istnet_pn2_set_tuning(1, conv) == 0
    try:
        xyz1 = torch.from_numpy(z["xyz_config1"]).to(DEV)
        fps = _ext.furthest_point_sampling(xyz1, 512)
        assert np.array_equal(fps.cpu().numpy(), z[f"c{conv}_config1_fps"].astype(np.int32))
        new_xyz = torch.gather(xyz1, 1, fps.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        assert np.array_equal(_ext.ball_query(new_xyz, xyz1, 0.2, 32).cpu().numpy(), z[f"c{conv}_config1_ball"].astype(np.int32))
        captured, counters = {}, {}

        def tap(name):
            orig = getattr(_ext, name)

            def fn(*a, **k):
                res = orig(*a, **k)
                i = counters.get(name, 0)
                counters[name] = i + 1
                captured[f"{name}_{i}"] = res
                return res
            monkeypatch.setattr(_ext, name, fn)
        for n in ("furthest_point_sampling_chain", "ball_query_pair", "three_nn_weights_multi"):
            tap(n)
        torch.manual_seed(0)
        enc = PointNet2MSG([list(r) for r in CAM]).to(DEV).train()
        pts = torch.from_numpy(z["pts_cube"]).to(DEV)
        with torch.no_grad():
            out = enc(pts)
        for lvl in range(4):
            captured[f"ball_query_{2 * lvl}"], captured[f"ball_query_{2 * lvl + 1}"] = captured[f"ball_query_pair_{lvl}"][:2]
            captured[f"three_nn_weights_{lvl}"] = captured["three_nn_weights_multi_0"][lvl]
        flips = 0
        for i in range(4):
            want = z[f"c{conv}_furthest_point_sampling_{i}"].astype(np.int32)
            assert np.array_equal(captured[f"furthest_point_sampling_chain_{i}"][0].cpu().numpy(), want), i
            flips += int((want != z[f"c0_furthest_point_sampling_{i}"]).sum())
        for i in range(8):
            want = z[f"c{conv}_ball_query_{i}"].astype(np.int32)
            assert np.array_equal(captured[f"ball_query_{i}"].cpu().numpy(), want), i
            flips += int((want != z[f"c0_ball_query_{i}"]).sum())
        levels = [pts.contiguous()] + [captured[f"furthest_point_sampling_chain_{k}"][1] for k in range(4)]
        for i in range(4):
            idx = captured[f"three_nn_weights_{i}"][0]
            assert np.array_equal(idx.cpu().numpy(), z[f"c{conv}_three_nn_idx_{i}"].astype(np.int32)), i
            d2, _ = _ext.three_nn(levels[3 - i].contiguous(), levels[4 - i].contiguous())
            assert np.array_equal(d2.cpu().numpy(), z[f"c{conv}_three_nn_dist2_{i}"]), i
        assert (flips > 0) == (conv != 0)            # the stored clouds do depend on the convention
        np.testing.assert_allclose(out.cpu().numpy()[:, :, ::8], z[f"c{conv}_out_train"], rtol=2e-4, atol=2e-4)
    finally:
        assert lib.istnet_pn2_set_tuning(1, 0) == 0


def test_istnet_point_branch_poses():
    """Predicted R / t / s and NOCS coordinates within 1e-4 of the reference composition."""
    from istnet_amd.ist_net import IST_Net
    z = np.load(os.path.join(GOLD, "istnet_point_branch_b2.npz"))
    torch.manual_seed(5)
    net = IST_Net()
    net.rgb_cam_extractor = torch.nn.Identity()
    net = net.to(DEV)
    b = 2
    inputs = {"rgb": torch.from_numpy(z["rgb_feat"]).to(DEV), "pts": torch.from_numpy(z["pts"]).to(DEV),
              "choose": torch.from_numpy(z["choose"].astype(np.int64)).to(DEV),
              "category_label": torch.from_numpy(z["cls"]).reshape(b, 1).to(DEV), "qo": torch.from_numpy(z["qo"]).to(DEV)}
    sub = lambda v: v.detach().cpu().numpy() if v.numel() <= 8192 else v.detach().cpu().numpy().reshape(b, -1)[:, ::64]
    net.train()
    ep = net(inputs)
    # train mode against the FLOAT64 evaluation of the reference composition with the same index decisions
    # (tests/golden/make_golden_f64.py; the reference's own fp32 run is within 7e-6 of it): every end point within 1e-4,
    # relative per element with an absolute floor of 1e-4 of the tensor's largest magnitude
    z64 = np.load(os.path.join(GOLD, "istnet_point_branch_f64.npz"))
    for k in [k[len("train_"):] for k in z.files if k.startswith("train_")]:
        want = z64["pb_" + k]
        np.testing.assert_allclose(sub(ep[k]), want, rtol=1e-4, atol=1e-4 * float(np.abs(want).max()), err_msg=k)
        np.testing.assert_allclose(sub(ep[k]), z["train_" + k], rtol=1e-4, atol=1e-4 * float(np.abs(want).max()), err_msg=k)
    net.eval()
    with torch.no_grad():
        ev = net(inputs)
    for k in ("pred_rotation", "pred_translation", "pred_size", "pred_qo"):
        np.testing.assert_allclose(sub(ev[k]), z["eval_" + k], **TOL, err_msg=k)


# ---------------------------------------------------------------------------------------------
# full-size (B=32, N=1024) size-independent properties
# ---------------------------------------------------------------------------------------------
def test_heads_match_reference_golden_on_gpu():
    """SURVEY 8c item 4 on the GPU: FeatureDeformer, LightEstimator, HeavyEstimator and Ortho6d2Mat against the fixture the
    reference's own modules produced (tests/golden/make_golden.py imports model/ist_net.py:114-332 and
    utils/rotation_utils.py:4-28 unmodified).  Same seed -> same initial weights (state checksum), outputs within the 1e-4
    bar; train mode, so the heads' BatchNorm batch statistics run through the native kernels."""
    from istnet_amd import ist_net, rotation_utils
    z = np.load(os.path.join(GOLD, "ist_heads_b2.npz"))
    t = lambda k: torch.from_numpy(z[k]).to(DEV)
    index = (t("cls").long() + torch.arange(2, device=DEV) * 6)
    cases = {"deformer": (ist_net.FeatureDeformer, (t("pts"), t("rgb_local"), t("pts_local"), index)),
             "light": (ist_net.LightEstimator, (t("pts"), t("rgb_local"), t("pts_local"))),
             "heavy": (ist_net.HeavyEstimator, (t("pts"), t("pts_w"), t("rgb_local"), t("pts_local"), t("pts_w_local")))}
    from istnet_amd.pointnet2 import fused_mlp
    fused_mlp.FALLBACKS.clear()
    for name, (ctor, args) in cases.items():
        torch.manual_seed(40)
        m = ctor()
        sd = {k: v for k, v in m.state_dict().items() if "running_" not in k and "num_batches" not in k}
        chk = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in sd.values()])
        np.testing.assert_allclose(chk, z[f"{name}_state_checksum"], rtol=0, atol=0)
        m = m.to(DEV)
        for i, o in enumerate(m(*args)):
            want = z[f"{name}_out{i}"]
            o = o.detach().cpu()
            got = o.numpy() if o.numel() < 4096 else o.numpy().reshape(o.shape[0], -1)[:, ::16]
            np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5, err_msg=f"{name}[{i}]")
    assert not fused_mlp.FALLBACKS, fused_mlp.FALLBACKS          # the native kernels ran, not the torch composition
    rot = rotation_utils.Ortho6d2Mat(t("x6"), t("y6")).cpu()
    np.testing.assert_allclose(rot.numpy(), z["rot"], rtol=1e-5, atol=1e-6)


def _shell(b, n, seed):
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(b, n, 3, generator=g)
    pts = d / d.norm(dim=2, keepdim=True) * 0.1 + torch.randn(b, n, 3, generator=g) * 0.002
    return (pts - pts.mean(dim=1, keepdim=True)).contiguous()


def test_full_size_index_properties(ext):
    b, n, m = 32, 1024, 512
    xyz = _shell(b, n, 0).to(DEV)
    fps = ext.furthest_point_sampling(xyz, m)
    assert (fps[:, 0] == 0).all() and int(fps.min()) >= 0 and int(fps.max()) < n
    assert all(len(set(row.tolist())) == m for row in fps.cpu())            # no repeats
    new_xyz = torch.gather(xyz, 1, fps.long().unsqueeze(-1).expand(b, m, 3)).contiguous()
    for r, ns in ((0.01, 16), (0.02, 32)):
        idx = ext.ball_query(new_xyz, xyz, r, ns).long()
        assert int(idx.min()) >= 0 and int(idx.max()) < n
        nb = torch.gather(xyz.unsqueeze(1).expand(b, m, n, 3), 2, idx.unsqueeze(-1).expand(b, m, ns, 3))
        d2 = ((nb - new_xyz.unsqueeze(2)) ** 2).sum(-1)
        assert bool((d2 < r * r * (1 + 1e-5)).all())                        # every slot inside the ball
        assert bool((idx[:, :, 0] <= idx.min(dim=2).values).all())          # first hit is the lowest index
        srt = idx.clone()
        # hits are strictly ascending until padding starts, padding repeats the first hit
        inc = idx[:, :, 1:] > idx[:, :, :-1]
        pad = idx[:, :, 1:] == idx[:, :, :1]
        assert bool((inc | pad).all())
    d2, idx3 = ext.three_nn(xyz, new_xyz)
    assert bool((d2[:, :, 0] <= d2[:, :, 1]).all() and (d2[:, :, 1] <= d2[:, :, 2]).all())
    # a centroid is its own nearest neighbour at distance 0
    self_d = torch.gather(d2[:, :, 0], 1, fps.long())
    assert float(self_d.max()) == 0.0


def test_full_size_group_and_interpolate_adjoint(ext):
    """<group(x), y> == <x, group_grad(y)> and the same for three_interpolate (linearity/adjoint)."""
    b, c, n, m, ns = 32, 64, 512, 256, 32
    g = torch.Generator().manual_seed(1)
    x = torch.randn(b, c, n, generator=g).to(DEV)
    idx = torch.randint(0, n, (b, m, ns), generator=g, dtype=torch.int32).to(DEV)
    y = torch.randn(b, c, m, ns, generator=g).to(DEV)
    lhs = (ext.group_points(x, idx).double() * y.double()).sum()
    rhs = (x.double() * ext.group_points_grad(y, idx, n).double()).sum()
    assert abs(float(lhs - rhs)) <= 1e-6 * max(1.0, abs(float(lhs))) + 1e-2
    idx3 = torch.randint(0, m, (b, n, 3), generator=g, dtype=torch.int32).to(DEV)
    w = torch.rand(b, n, 3, generator=g).to(DEV)
    f = torch.randn(b, c, m, generator=g).to(DEV)
    yo = torch.randn(b, c, n, generator=g).to(DEV)
    lhs = (ext.three_interpolate(f, idx3, w).double() * yo.double()).sum()
    rhs = (f.double() * ext.three_interpolate_grad(yo, idx3, w, m).double()).sum()
    assert abs(float(lhs - rhs)) <= 1e-6 * max(1.0, abs(float(lhs))) + 1e-2


class _F64Ext:
    """`_ext` stand-in for a float64 evaluation of the encoder on the GPU: index decisions come from the real
    (bit-exact) fp32 kernels, every feature-valued op is plain torch in float64.  The SharedMLPs of a .double()
    model are not fusable, so the whole dense part is torch float64 as well -- an arithmetic 'truth' to which
    both fp32 implementations (HIP path, CPU oracle composition) are compared."""

    def __init__(self, ext):
        self.ext = ext

    def furthest_point_sampling(self, xyz, m):
        return self.ext.furthest_point_sampling(xyz.float().contiguous(), m)

    def ball_query(self, new_xyz, xyz, radius, nsample):
        return self.ext.ball_query(new_xyz.float().contiguous(), xyz.float().contiguous(), radius, nsample)

    def three_nn(self, unknown, known):
        d2, idx = self.ext.three_nn(unknown.float().contiguous(), known.float().contiguous())
        return d2.double(), idx

    @staticmethod
    def gather_points(points, idx):
        return torch.gather(points, 2, idx.long().unsqueeze(1).expand(-1, points.size(1), -1))

    @staticmethod
    def group_points(points, idx):
        b, c, n = points.shape
        flat = idx.long().reshape(b, 1, -1).expand(-1, c, -1)
        return torch.gather(points, 2, flat).reshape(b, c, idx.size(1), idx.size(2))

    @staticmethod
    def three_interpolate(points, idx, weight):
        b, c, m = points.shape
        n = idx.size(1)
        taps = torch.gather(points, 2, idx.long().reshape(b, 1, -1).expand(-1, c, -1)).reshape(b, c, n, 3)
        return (taps * weight.unsqueeze(1)).sum(dim=3)

    # gradients of the three differentiable ops (the autograd Functions of pointnet2_utils call them in backward)
    @staticmethod
    def gather_points_grad(grad_out, idx, n):
        b, c, m = grad_out.shape
        return torch.zeros(b, c, n, dtype=grad_out.dtype, device=grad_out.device).scatter_add_(
            2, idx.long().unsqueeze(1).expand(-1, c, -1), grad_out)

    @staticmethod
    def group_points_grad(grad_out, idx, n):
        b, c = grad_out.shape[:2]
        flat = idx.long().reshape(b, 1, -1).expand(-1, c, -1)
        return torch.zeros(b, c, n, dtype=grad_out.dtype, device=grad_out.device).scatter_add_(2, flat, grad_out.reshape(b, c, -1))

    @staticmethod
    def three_interpolate_grad(grad_out, idx, weight, m, csr=None):
        b, c, n = grad_out.shape
        contrib = (grad_out.unsqueeze(3) * weight.unsqueeze(1)).reshape(b, c, 3 * n)
        return torch.zeros(b, c, m, dtype=grad_out.dtype, device=grad_out.device).scatter_add_(
            2, idx.long().reshape(b, 1, -1).expand(-1, c, -1), contrib)


def test_full_size_encoder_matches_cpu_oracle_composition(oracle, ext):
    """B=32 N=1024 encoder forward: the HIP path and the CPU oracle composition (both fp32) against a float64
    evaluation with the same index decisions.  Each fp32 implementation must be within 1e-4 of the float64
    result; the two fp32 results then differ by at most the sum of their errors."""
    from istnet_amd.modules import PointNet2MSG
    from istnet_amd.pointnet2 import pointnet2_utils
    torch.manual_seed(0)
    enc = PointNet2MSG([list(r) for r in CAM]).train()
    pts = _shell(32, 1024, 0)
    enc_gpu = PointNet2MSG([list(r) for r in CAM]).to(DEV).train()
    enc_gpu.load_state_dict(enc.state_dict())
    enc64 = PointNet2MSG([list(r) for r in CAM]).to(DEV).double().train()
    enc64.load_state_dict(enc.state_dict())
    out_gpu = enc_gpu(pts.to(DEV)).detach().cpu()
    saved = pointnet2_utils._ext
    try:
        pointnet2_utils._ext = _F64Ext(ext)
        out64 = enc64(pts.to(DEV).double()).detach().cpu()
        pointnet2_utils._ext = oracle
        out_cpu = enc(pts).detach()
    finally:
        pointnet2_utils._ext = saved
    torch.testing.assert_close(out_gpu.double(), out64, **TOL)
    torch.testing.assert_close(out_cpu.double(), out64, **TOL)
    torch.testing.assert_close(out_gpu, out_cpu, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("kind", ["cube", "dense"])
def test_full_size_encoder_on_the_other_input_distributions(kind, oracle, ext):
    """SURVEY 8d names two input distributions for config 2 (cube and shell); ``dense`` is the case where every ball is
    over-full (no padded rows: nothing for the compact columns or the first-hit padding to save).  B=32 N=1024: every index
    tensor of the geometry pass bit-exact against the oracle (FPS picks, both ball queries, three_nn of every level), and the
    features of the HIP path and of the CPU oracle composition within 1e-4 of a float64 evaluation."""
    import bench
    from istnet_amd.modules import PointNet2MSG
    from istnet_amd.pointnet2 import pointnet2_utils
    pts = bench.CLOUDS[kind](32, 1024, seed=0)
    # ---- indices, level by level, on the oracle's own picks (a mismatch is then local to the op that made it) ----
    cur = pts
    levels = [pts]
    for (npoint, radii) in zip((512, 256, 128, 64), CAM):
        want = oracle.furthest_point_sampling(cur, npoint)
        got = ext.furthest_point_sampling(cur.to(DEV), npoint).cpu()
        assert torch.equal(got, want), (kind, npoint, "fps")
        new_xyz = torch.gather(cur, 1, want.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        for radius, nsample in zip(radii, (16, 32)):
            bw = oracle.ball_query(new_xyz, cur, radius, nsample)
            bg = ext.ball_query(new_xyz.to(DEV), cur.to(DEV), radius, nsample).cpu()
            assert torch.equal(bg, bw), (kind, npoint, radius)
            if kind == "dense":
                assert (bw[..., 1:] != bw[..., :1]).any(-1).all()        # no row is a single padded hit
        levels.append(new_xyz)
        cur = new_xyz
    for unknown, known in zip(levels[:-1], levels[1:]):
        dw, iw = oracle.three_nn(unknown, known)
        dg, ig = ext.three_nn(unknown.to(DEV), known.to(DEV))
        assert torch.equal(ig.cpu(), iw) and torch.equal(dg.cpu(), dw), (kind, unknown.shape[1])
    # ---- features ----
    torch.manual_seed(0)
    enc = PointNet2MSG([list(r) for r in CAM]).train()
    enc_gpu = PointNet2MSG([list(r) for r in CAM]).to(DEV).train()
    enc_gpu.load_state_dict(enc.state_dict())
    enc64 = PointNet2MSG([list(r) for r in CAM]).to(DEV).double().train()
    enc64.load_state_dict(enc.state_dict())
    out_gpu = enc_gpu(pts.to(DEV)).detach().cpu()
    saved = pointnet2_utils._ext
    try:
        pointnet2_utils._ext = _F64Ext(ext)
        out64 = enc64(pts.to(DEV).double()).detach().cpu()
        pointnet2_utils._ext = oracle
        out_cpu = enc(pts).detach()
    finally:
        pointnet2_utils._ext = saved
    torch.testing.assert_close(out_gpu.double(), out64, **TOL)          # the product: the 1e-4 bar
    # the CPU composition is torch's own fp32 layers (the reference's arithmetic): on the dense cloud ONE element of 4.2 M
    # ends 1.22e-4 from float64 (measured, round 6), like the 1.4e-4 / 2.1e-4 the reference composition shows elsewhere (below)
    torch.testing.assert_close(out_cpu.double(), out64, rtol=2.5e-4, atol=2.5e-4)


def test_compact_columns_follow_the_data(monkeypatch):
    """fused_mlp.COMPACT_POLICY "auto" (round 6): the first two passes of a model read the compact-column fill ratio back; shell
    and cube clouds keep the compact columns of level 1, a dense cloud (every ball over-full) drops them -- and the step computes
    the same thing either way (compact columns are an exact re-grouping of the padded sums: fp32 round-off apart)."""
    import bench
    from istnet_amd.modules import PointNet2MSG
    from istnet_amd.pointnet2 import fused_mlp

    def run(kind, policy, steps=3):
        monkeypatch.setattr(fused_mlp, "COMPACT_POLICY", policy)
        torch.manual_seed(0)
        enc = PointNet2MSG([list(r) for r in CAM]).to(DEV).train()
        pts = bench.CLOUDS[kind](8, 1024, seed=3, device=DEV)
        for _ in range(steps):
            enc.zero_grad(set_to_none=True)
            out = enc(pts)
            out.square().mean().backward()
        torch.cuda.synchronize()
        return enc, out.detach(), [p.grad.clone() for p in enc.parameters()]

    enc, out_auto, g_auto = run("dense", "auto")
    assert enc._compact_off == {0: True} and min(enc._compact_fill[0]) > 0.95
    _, out_on, g_on = run("dense", "on")
    # two fp32 evaluations of the same sums in different orders (BatchNorm statistics included), each within 1e-4 of float64
    # (test_full_size_encoder_on_the_other_input_distributions): they differ by at most the sum -- measured 1.2e-4
    torch.testing.assert_close(out_auto, out_on, rtol=2.5e-4, atol=2.5e-4)
    # gradients: a max-pool arg-max that resolves differently under the two summation orders re-routes a gradient column, so
    # two fp32 evaluations agree in the aggregate only (the yardstick of test_encoder_parameter_gradients_*: ~5e-3 relative L2)
    num = sum(float((a - b_).pow(2).sum()) for a, b_ in zip(g_auto, g_on)) ** 0.5
    den = sum(float(b_.pow(2).sum()) for b_ in g_on) ** 0.5
    assert num / den < 2e-2, num / den
    for kind in ("shell", "cube"):
        enc, _, _ = run(kind, "auto", steps=2)
        assert enc._compact_off == {0: False} and max(enc._compact_fill[0]) < 0.5, (kind, enc._compact_fill)
    enc, _, _ = run("dense", "on", steps=2)
    assert "_compact_off" not in enc.__dict__ or enc._compact_off == {}


def test_inference_config5_n2048_matches_cpu_oracle_composition(oracle):
    """BASELINE config 5 shape (eval mode, N=2048; B reduced so the CPU side stays fast): the IST-Net point
    branch on the GPU vs the same modules over the CPU oracle; poses within 1e-4."""
    from istnet_amd.ist_net import IST_Net
    from istnet_amd.pointnet2 import pointnet2_utils
    b, n = 4, 2048
    torch.manual_seed(11)
    net = IST_Net().eval()
    g = torch.Generator().manual_seed(12)
    pts = _shell(b, n, 13) + torch.tensor([0.0, 0.0, 0.8])
    inputs = {"pts": pts, "rgb_local": torch.randn(b, 128, n, generator=g),
              "category_label": torch.randint(0, 6, (b, 1), generator=g)}
    net_gpu = IST_Net().to(DEV).eval()
    net_gpu.load_state_dict(net.state_dict())
    with torch.no_grad():
        out_gpu = net_gpu({k: v.to(DEV) for k, v in inputs.items()})
        saved = pointnet2_utils._ext
        try:
            pointnet2_utils._ext = oracle
            out_cpu = net(inputs)
        finally:
            pointnet2_utils._ext = saved
    for k in ("pred_rotation", "pred_translation", "pred_size", "pred_qo"):
        torch.testing.assert_close(out_gpu[k].cpu(), out_cpu[k], **TOL)
    # rotations are proper
    r = out_gpu["pred_rotation"]
    torch.testing.assert_close(torch.matmul(r.transpose(1, 2), r), torch.eye(3, device=DEV).expand(b, 3, 3),
                               rtol=1e-4, atol=1e-4)
    # the same deltas in the evaluation's own units (degrees / cm, evaluation_utils.py:632-660) after the
    # post-processing of test_func (solver.py:231-241); class ids without symmetry so the full rotation counts
    from istnet_amd import postprocess
    rts_gpu, scales_gpu = postprocess.assemble_pred_RTs(*(out_gpu[k].cpu() for k in ("pred_rotation", "pred_translation", "pred_size")))
    rts_cpu, scales_cpu = postprocess.assemble_pred_RTs(*(out_cpu[k] for k in ("pred_rotation", "pred_translation", "pred_size")))
    err = postprocess.pose_errors(rts_gpu, rts_cpu, [3] * b, [1] * b).diagonal(dim1=0, dim2=1)
    assert float(err[0].max()) < 0.05 and float(err[1].max()) < 0.01, err      # degrees, centimetres
    torch.testing.assert_close(scales_gpu, scales_cpu, **TOL)


def test_full_istnet_with_rgb_branch_trains_one_step():
    """End-to-end wiring (config 3 shape, tiny): RGB branch on MIOpen + point branch on the HIP kernels,
    SupervisedLoss, backward, every parameter receives a finite gradient."""
    from istnet_amd.ist_net import IST_Net
    from istnet_amd.losses import SupervisedLoss
    from istnet_amd.rgb_branch import ModifiedResnet
    torch.manual_seed(3)
    b, n, hw = 2, 1024, 96
    net = IST_Net(rgb_extractor=ModifiedResnet()).to(DEV).train()
    g = torch.Generator().manual_seed(4)
    inputs = {"rgb": torch.randn(b, 3, hw, hw, generator=g).to(DEV),
              "pts": (_shell(b, n, 5) + torch.tensor([0.0, 0.0, 0.8])).to(DEV),
              "choose": torch.randint(0, hw * hw, (b, n), generator=g).to(DEV),
              "category_label": torch.randint(0, 6, (b, 1), generator=g).to(DEV),
              "qo": (torch.rand(b, n, 3, generator=g) - 0.5).to(DEV)}
    ep = net(inputs)
    ep.update({"rotation_label": torch.linalg.qr(torch.randn(b, 3, 3, generator=g))[0].to(DEV),
               "translation_label": inputs["pts"].mean(dim=1), "size_label": torch.rand(b, 3, generator=g).to(DEV) * 0.25 + 0.05,
               "qo": inputs["qo"]})
    loss = SupervisedLoss(1.0, 10.0)(ep)
    assert torch.isfinite(loss)
    loss.backward()
    missing = [k for k, p in net.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
    # avgpool/fc of the trunk are unused by design (reference keeps them for the checkpoint layout)
    assert [k for k in missing if ".fc." not in k] == []
    assert len(net.state_dict()) > 400


def test_benchmark_path_runs_only_native_kernels_for_the_dense_stack():
    """On the GPU the encoder must not fall back to torch Conv2d / BatchNorm2d / max_pool2d: the dense stack
    runs in the MFMA kernels of libistnet_pn2.so (module forwards are bypassed), and the library is loaded."""
    from istnet_amd import _native
    from istnet_amd.modules import PointNet2MSG
    torch.manual_seed(0)
    enc = PointNet2MSG([list(r) for r in CAM]).to(DEV).train()
    fired, handles = [], []
    for name, m in enc.named_modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.BatchNorm2d, torch.nn.ReLU)):
            handles.append(m.register_forward_hook(lambda mod, i, o, name=name: fired.append(name)))
    try:
        out = enc(_shell(4, 1024, 1).to(DEV))
        out.square().mean().backward()
    finally:
        # the activation module is ONE instance shared by every stack of the process (a default argument, as in the
        # reference's pytorch_utils.py:25-50): a hook left on it follows every model built afterwards -- and switches their
        # graph segments off (graphed.WHY says so)
        for h in handles:
            h.remove()
    assert fired == [], f"torch fallback used for {fired[:4]}"
    assert _native._lib is not None and os.path.exists(_native.LIB_PATH)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in enc.parameters())


@pytest.mark.parametrize("b,n,extra", [(1, 1024, 0), (3, 1000, 0), (2, 768, 3), (2, 1024, 3)])
def test_encoder_odd_shapes_match_cpu_oracle_composition(oracle, b, n, extra):
    """Shapes off the benchmark path: batch 1, N not a multiple of 32 (the fused FP node declines, the reference
    composition runs), input clouds with extra per-point features (level 1 then has features).  Forward within
    1e-4 of the CPU oracle composition; backward runs and gives finite gradients of the right norm."""
    from istnet_amd.modules import PointNet2MSG
    from istnet_amd.pointnet2 import pointnet2_utils
    torch.manual_seed(b * 1000 + n + extra)
    enc = PointNet2MSG([list(r) for r in CAM]).train()
    if extra:   # rebuild level 1 for extra input channels, as a caller with coloured clouds would
        from istnet_amd.pointnet2.pointnet2_modules import PointnetSAModuleMSG
        enc.SA_modules[0] = PointnetSAModuleMSG(npoint=512, radii=list(CAM[0]), nsamples=[16, 32],
                                                mlps=[[extra, 16, 16, 32], [extra, 16, 16, 32]], use_xyz=True, bn=True)
        enc.FP_modules[0] = __import__("istnet_amd").pointnet2.pointnet2_modules.PointnetFPModule(
            mlp=[256 + extra, 128, 128], bn=True)
    g = torch.Generator().manual_seed(n)
    pts = torch.cat([_shell(b, n, n + b), torch.randn(b, n, extra, generator=g)], dim=2) if extra else _shell(b, n, n + b)
    import copy
    enc_gpu = copy.deepcopy(enc).to(DEV).train()
    out_gpu = enc_gpu(pts.to(DEV))
    saved = pointnet2_utils._ext
    try:
        pointnet2_utils._ext = oracle
        out_cpu = enc(pts)
        out_cpu.square().mean().backward()
    finally:
        pointnet2_utils._ext = saved
    torch.testing.assert_close(out_gpu.detach().cpu(), out_cpu.detach(), rtol=2e-4, atol=2e-4)
    out_gpu.square().mean().backward()
    for (name, p), q in zip(enc_gpu.named_parameters(), enc.parameters()):
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
    gn = torch.stack([p.grad.norm() for p in enc_gpu.parameters()]).cpu()
    cn = torch.stack([q.grad.norm() for q in enc.parameters()])
    assert float(((gn - cn).abs() / (cn + 1e-12)).median()) < 2e-2


def test_eval_rgb_tail_on_chosen_pixels_matches_dense_order():
    """Eval mode on the GPU (channels-last extractor, as bench.py builds it): IST-Net poses with the RGB decoder tail
    evaluated at the chosen pixels only == the reference order (dense feature map, then the `choose` gather)."""
    from istnet_amd import ist_net, rgb_branch
    torch.manual_seed(21)
    ext = rgb_branch.ModifiedResnet()
    g = torch.Generator().manual_seed(22)
    for m in ext.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
    net = ist_net.IST_Net(rgb_extractor=ext).to(DEV).eval()
    net.rgb_cam_extractor.to(memory_format=torch.channels_last)
    b, n, hw = 2, 256, 96
    inputs = {"rgb": torch.randn(b, 3, hw, hw, generator=g).to(DEV).contiguous(memory_format=torch.channels_last),
              "pts": (_shell(b, n, 23) + torch.tensor([0.0, 0.0, 0.8])).to(DEV),
              "choose": torch.randint(0, hw * hw, (b, n), generator=g).to(DEV),
              "category_label": torch.randint(0, 6, (b, 1), generator=g).to(DEV)}
    inputs["choose"][0, :4] = torch.tensor([0, hw - 1, hw * (hw - 1), hw * hw - 1], device=DEV)   # image corners
    saved = ist_net.USE_GATHER_FIRST
    try:
        with torch.no_grad():
            ist_net.USE_GATHER_FIRST = True
            fast = net(inputs)
            ist_net.USE_GATHER_FIRST = False
            dense = net(inputs)
    finally:
        ist_net.USE_GATHER_FIRST = saved
    for k in ("pred_rotation", "pred_translation", "pred_size", "pred_qo"):
        torch.testing.assert_close(fast[k], dense[k], rtol=1e-4, atol=1e-5)


def test_istnet_frozen_world_enhancer_on_gpu():
    """Second training stage (train.py:102-118): world enhancer frozen.  End points and loss within tolerance of the
    reference's golden values on the HIP path; the frozen encoder receives no gradient, everything else does."""
    from istnet_amd import losses
    from istnet_amd.ist_net import IST_Net
    z = np.load(os.path.join(GOLD, "istnet_freeze_b2.npz"))
    torch.manual_seed(7)
    net = IST_Net(freeze_world_enhancer=True)
    net.rgb_cam_extractor = torch.nn.Identity()
    net = net.to(DEV).train()
    for p in net.world_enhancer.parameters():          # train.py:116-118 leaves these out of the optimizer
        p.requires_grad_(False)
    b = 2
    inputs = {"rgb": torch.from_numpy(z["rgb_feat"]).to(DEV), "pts": torch.from_numpy(z["pts"]).to(DEV),
              "choose": torch.from_numpy(z["choose"].astype(np.int64)).to(DEV),
              "category_label": torch.from_numpy(z["cls"]).reshape(b, 1).to(DEV), "qo": torch.from_numpy(z["qo"]).to(DEV)}
    labels = {k[4:]: torch.from_numpy(z[k]).to(DEV) for k in z.files if k.startswith("lab_")}
    ep = net(inputs)
    sub = lambda v: v.detach().cpu().numpy() if v.numel() <= 8192 else v.detach().cpu().numpy().reshape(b, -1)[:, ::64]
    keys = [k[len("train_"):] for k in z.files if k.startswith("train_")]
    assert set(keys) == set(ep.keys())
    z64 = np.load(os.path.join(GOLD, "istnet_point_branch_f64.npz"))       # float64 evaluation of the reference (make_golden_f64.py)
    for k in keys:
        want = z64["fz_" + k]
        np.testing.assert_allclose(sub(ep[k]), want, rtol=1e-4, atol=1e-4 * float(np.abs(want).max()), err_msg=k)
    loss = losses.SupervisedLoss(1.0, 10.0, freeze_world_enhancer=True)({**ep, **labels, "qo": inputs["qo"]})
    np.testing.assert_allclose(float(loss.detach()), float(z["loss"]), rtol=1e-4)
    np.testing.assert_allclose(float(loss.detach()), float(z64["fz_loss"]), rtol=2e-5)
    loss.backward()
    assert all(p.grad is None for p in net.world_enhancer.parameters())
    missing = [n for n, p in net.named_parameters() if p.requires_grad and p.grad is None]
    assert not missing, missing
    assert all(bool(torch.isfinite(p.grad).all()) for p in net.parameters() if p.grad is not None)


def test_rgb_branch_matches_reference_golden_on_gpu():
    """the RGB branch (ResNet-18 trunk + PSP decoder on MIOpen) against the feature map the
    reference's own modules produced (tests/golden/rgb_branch_and_loss.npz), in both memory layouts the product uses:
    NCHW and channels-last (the bench's layout), and through the eval-mode tail that evaluates the decoder's last stage
    on the chosen pixels only.  Features within 1e-4."""
    from istnet_amd import rgb_branch
    from istnet_amd.ist_net import IST_Net
    z = np.load(os.path.join(GOLD, "rgb_branch_and_loss.npz"))
    torch.manual_seed(60)
    rgb_branch.ResNet()                      # the reference consumed one trunk's worth of the random stream first
    net = rgb_branch.ModifiedResnet().eval().to(DEV)
    img = torch.from_numpy(z["img"]).to(DEV)
    with torch.no_grad():
        out = net(img)
        net_cl = net.to(memory_format=torch.channels_last)
        out_cl = net_cl(img.contiguous(memory_format=torch.channels_last))
    assert out.shape == (1, 128, 96, 96)
    np.testing.assert_allclose(out.cpu().numpy()[:, ::4, ::6, ::6], z["out_sub"], **TOL)
    np.testing.assert_allclose(out_cl.cpu().numpy()[:, ::4, ::6, ::6], z["out_sub"], **TOL)
    # the per-point gather of IST_Net (dense order in train mode, tail-at-chosen-pixels in eval mode) picks these values
    g = torch.Generator().manual_seed(1)
    choose = torch.randint(0, 96 * 96, (1, 512), generator=g).to(DEV)
    model = IST_Net(rgb_extractor=net_cl).to(DEV).eval()
    with torch.no_grad():
        local = model._rgb_local({"rgb": img.contiguous(memory_format=torch.channels_last), "choose": choose}, 1)
    want = out.reshape(1, 128, -1).gather(2, choose.unsqueeze(1).expand(-1, 128, -1))
    torch.testing.assert_close(local, want, **TOL)
    sub_rows = (torch.arange(0, 96, 6).view(-1, 1) * 96 + torch.arange(0, 96, 6).view(1, -1)).reshape(1, -1).to(DEV)
    with torch.no_grad():
        local_sub = model._rgb_local({"rgb": img.contiguous(memory_format=torch.channels_last), "choose": sub_rows}, 1)
    np.testing.assert_allclose(local_sub.cpu().numpy()[:, ::4].reshape(1, 32, 16, 16), z["out_sub"], **TOL)


def test_encoder_parameter_gradients_vs_float64_autograd(ext):
    """Every parameter gradient of the full encoder (train-mode BN, loss = mean(out^2), the bench's shell clouds, B=8)
    against a float64 autograd evaluation of the same model with the same index decisions (`_F64Ext`: index ops from the
    bit-exact kernels, every feature op and the whole dense stack in float64).  Relative L2 error per tensor and over all
    1.3 M parameters together; the torch-fp32 composition (fused path off) is measured on the same footing and is the
    yardstick (see the comment at the assertions: 1e-4 against float64 is not attainable by any fp32 evaluation)."""
    from istnet_amd.modules import PointNet2MSG
    from istnet_amd.pointnet2 import fused_mlp, pointnet2_utils
    torch.manual_seed(0)
    base = PointNet2MSG([list(r) for r in CAM]).train()
    state = {k: v.clone() for k, v in base.state_dict().items()}
    pts = _shell(8, 1024, 0).to(DEV)

    def grads(double=False, fused=True):
        m = PointNet2MSG([list(r) for r in CAM]).to(DEV).train()
        m.load_state_dict(state)
        saved_ext, saved_fus = pointnet2_utils._ext, fused_mlp._fusable_shape
        try:
            if double:
                m = m.double()
                pointnet2_utils._ext = _F64Ext(ext)
            if not fused:
                fused_mlp._fusable_shape = lambda *a, **k: False
            out = m(pts.double() if double else pts)
            out.square().mean().backward()
        finally:
            pointnet2_utils._ext, fused_mlp._fusable_shape = saved_ext, saved_fus
        torch.cuda.synchronize()
        return {k: p.grad.double() for k, p in m.named_parameters()}

    g64, ghip, gtorch = grads(double=True), grads(), grads(fused=False)

    def rel(g):
        per = {k: float((g[k] - g64[k]).norm() / g64[k].norm().clamp_min(1e-30)) for k in g64}
        num = sum(float((g[k] - g64[k]).pow(2).sum()) for k in g64) ** 0.5
        den = sum(float(g64[k].pow(2).sum()) for k in g64) ** 0.5
        return per, num / den

    per_hip, all_hip = rel(ghip)
    per_torch, all_torch = rel(gtorch)
    # Measured (profiles/r02_gradient_budget.txt): BOTH fp32 evaluations sit ~5e-3 (relative L2, all parameters) from the
    # float64 gradient -- the torch composition, i.e. the reference's own arithmetic, 4.7e-3, the fused path 6.8e-3:
    # a max-pool arg-max that resolves differently in fp32 and float64 re-routes a whole gradient column, so the
    # gradient is a discontinuous function of round-off and no fp32 implementation can meet 1e-4 against float64.  The
    # bar that CAN be held is "as close to float64 as the reference's fp32 arithmetic": aggregate error within 2.5x of the
    # torch composition's, every tensor of non-negligible norm within 4x of it.  (Layer-level gradients, where no
    # re-routing happens, are checked at ~1e-6 against float64 in test_fused_mlp_gpu.py::test_fused_vs_float64.)
    top = max(float(v.norm()) for v in g64.values())
    big = [k for k in g64 if float(g64[k].norm()) > 1e-6 * top]
    worst = max(big, key=lambda k: per_hip[k] / (per_torch[k] + 1e-4))
    msg = (f"all parameters: hip {all_hip:.2e}, torch-fp32 {all_torch:.2e}; worst tensor vs torch {worst}: hip {per_hip[worst]:.2e}, "
           f"torch-fp32 {per_torch[worst]:.2e}; {len(big)} of {len(g64)} tensors above the norm floor")
    print(msg)
    # (the torch composition's own error moves from run to run -- its scatter-adds use atomics, and one re-routed arg-max
    # changes a tensor's error by a factor -- so the factors carry a margin over the measured 1.3x / 2.4x)
    assert all_hip < 2.5 * all_torch + 1e-5, msg
    assert all_hip < 2e-2 and all_torch < 2e-2, msg
    for k in big:
        assert per_hip[k] < 4.0 * per_torch[k] + 2e-3, f"{k}: hip {per_hip[k]:.2e} torch {per_torch[k]:.2e} | {msg}"


@pytest.mark.parametrize("which", ["golden", "shell32"])
def test_encoder_error_budget_per_level(which):
    """The 1e-4 bar, budgeted per level (tools/error_budget.py; tables in profiles/r02_error_budget_*.txt): every SA / FP
    level, fed the float64 evaluation's inputs, is within 3e-5 of the float64 output (fp32 round-off of one level, 3x
    inside the bar); what the end-to-end output shows beyond that is amplification through 16 normalised layers, and at
    every level the fused path's accumulated error stays at or below that of the torch-fp32 composition -- the
    reference's own arithmetic -- which itself ends 1.4e-4 (B=2) / 2.1e-4 (B=32) from float64."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import error_budget
    rows, final, _ = error_budget.compute(which)
    assert [r["level"] for r in rows] == ["SA1", "SA2", "SA3", "SA4", "FP4", "FP3", "FP2", "FP1"]
    for r in rows:
        assert r["hip_loc_max"] < 3e-5, r                              # per level: 1e-4 with a 3x margin
        assert r["hip_cum_max"] < 1.25 * r["torch_cum_max"] + 2e-6, r  # never behind the reference's fp32 arithmetic
        assert r["hip_cum_rms"] < 1.25 * r["torch_cum_rms"] + 2e-7, r
    assert final["hip"][0] < 2e-4 and final["hip"][1] < 1e-4


def test_rgb_branch_train_mode_golden_on_gpu():
    """Training-mode golden of the RGB branch (tests/golden/make_golden_rgb_train.py: the reference's ModifiedResnet with
    batch-statistics BatchNorm, the two Dropout2d layers in eval): dense output in both memory layouts and the
    training-mode tail that evaluates `final` at the chosen pixels with exact batch statistics, all within 1e-4 of the
    float64 evaluation of the reference; the last BatchNorm's running statistics after the step."""
    from istnet_amd import rgb_branch
    z = np.load(os.path.join(GOLD, "rgb_branch_train.npz"))

    def build():
        torch.manual_seed(60)
        rgb_branch.ResNet()                      # the reference consumed one trunk's worth of the random stream first
        net = rgb_branch.ModifiedResnet()
        g = torch.Generator().manual_seed(int(z["bn_seed"]))
        with torch.no_grad():
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
        net.train()
        for m in net.modules():
            if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
                m.eval()
        return net.to(DEV)

    img = torch.from_numpy(z["img"]).to(DEV)
    want = z["out_sub_f64"]
    tol = dict(rtol=1e-4, atol=1e-4 * float(np.abs(want).max()))
    net = build()
    out = net(img)
    np.testing.assert_allclose(out.detach().cpu().numpy()[:, ::4, ::6, ::6], want, **tol)
    np.testing.assert_allclose(out.detach().cpu().numpy()[:, ::4, ::6, ::6], z["out_sub"], **tol)
    last_bn = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)][-1]
    np.testing.assert_allclose(last_bn.running_mean.cpu().numpy(), z["last_running_mean"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(last_bn.running_var.cpu().numpy(), z["last_running_var"], rtol=1e-4, atol=1e-7)
    # channels-last (the bench's layout) + the training-mode tail at the chosen pixels: the sub-sampled grid as `choose`
    net_cl = build().to(memory_format=torch.channels_last)
    sub_rows = (torch.arange(0, 96, 6).view(-1, 1) * 96 + torch.arange(0, 96, 6).view(1, -1)).reshape(1, -1).expand(2, -1)
    local = net_cl(img.contiguous(memory_format=torch.channels_last), sub_rows.contiguous().to(DEV))
    assert local.shape == (2, 128, 256)
    np.testing.assert_allclose(local.detach().cpu().numpy()[:, ::4].reshape(2, 32, 16, 16), want, **tol)
    last_cl = [m for m in net_cl.modules() if isinstance(m, torch.nn.BatchNorm2d)][-1]
    np.testing.assert_allclose(last_cl.running_var.cpu().numpy(), z["last_running_var"], rtol=1e-4, atol=1e-7)
    local.square().mean().backward()
    assert all(torch.isfinite(p.grad).all() for n, p in net_cl.named_parameters() if p.grad is not None)


def test_supervised_loss_golden_on_gpu():
    """SupervisedLoss (model/ist_net.py:78-111) of the reference on the train-mode end points of the point branch: the
    value the reference's own loss module produced (rgb_branch_and_loss.npz) from OUR end points on the GPU -- the
    native PoseDis / SmoothL1Dis nodes and the full feature maps included -- within 1e-4, and a finite gradient everywhere."""
    from istnet_amd.ist_net import IST_Net
    from istnet_amd import losses
    z = np.load(os.path.join(GOLD, "istnet_point_branch_b2.npz"))
    zl = np.load(os.path.join(GOLD, "rgb_branch_and_loss.npz"))
    torch.manual_seed(5)
    net = IST_Net()
    net.rgb_cam_extractor = torch.nn.Identity()
    net = net.to(DEV).train()
    b = 2
    inputs = {"rgb": torch.from_numpy(z["rgb_feat"]).to(DEV), "pts": torch.from_numpy(z["pts"]).to(DEV),
              "choose": torch.from_numpy(z["choose"].astype(np.int64)).to(DEV),
              "category_label": torch.from_numpy(z["cls"]).reshape(b, 1).to(DEV), "qo": torch.from_numpy(z["qo"]).to(DEV)}
    ep = net(inputs)
    ep.update({k[4:]: torch.from_numpy(zl[k]).to(DEV) for k in zl.files if k.startswith("lab_")})
    ep["qo"] = inputs["qo"]
    loss = losses.SupervisedLoss(1.0, 10.0, False)(ep)
    np.testing.assert_allclose(float(loss.detach()), float(zl["loss"]), rtol=1e-4)
    loss.backward()
    bad = [n for n, p in net.named_parameters() if p.grad is None or not bool(torch.isfinite(p.grad).all())]
    assert not bad, bad


def test_supervised_loss_gradients_match_the_written_chain_in_float64():
    """SupervisedLoss on the GPU (native PoseDis / SmoothL1Dis, one-pass MSE, the weighted sum as one node) against the
    reference's chain of adds and multiplies (model/ist_net.py:95-110, model/losses.py:3-49) evaluated in float64 on the
    host: value and the gradient of every prediction, the feature target included (both sides of the MSE carry a gradient)."""
    from istnet_amd import losses
    g = torch.Generator().manual_seed(11)
    b, n = 6, 128

    def rot():
        q, _ = torch.linalg.qr(torch.randn(b, 3, 3, generator=g))
        return q

    names = {"pred_rotation": rot(), "pred_translation": torch.randn(b, 3, generator=g), "pred_size": torch.rand(b, 3, generator=g),
             "pred_rotation_aux_cam": rot(), "pred_translation_aux_cam": torch.randn(b, 3, generator=g),
             "pred_size_aux_cam": torch.rand(b, 3, generator=g), "pred_rotation_aux_world": rot(),
             "pred_translation_aux_world": torch.randn(b, 3, generator=g), "pred_size_aux_world": torch.rand(b, 3, generator=g),
             "pred_qo": torch.randn(b, n, 3, generator=g) * 0.2, "pts_w_local": torch.randn(b, 64, n, generator=g),
             "pts_w_local_gt": torch.randn(b, 64, n, generator=g)}
    fixed = {"rotation_label": rot(), "translation_label": torch.randn(b, 3, generator=g), "size_label": torch.rand(b, 3, generator=g),
             "qo": torch.randn(b, n, 3, generator=g) * 0.2}

    def run(dev, dtype):
        ep = {k: v.to(device=dev, dtype=dtype).requires_grad_(True) for k, v in names.items()}
        ep.update({k: v.to(device=dev, dtype=dtype) for k, v in fixed.items()})
        loss = losses.SupervisedLoss(1.0, 10.0, False)(ep)
        loss.backward()
        return float(loss.detach()), {k: ep[k].grad.detach().cpu().double() for k in names}

    lg, gg = run(DEV, torch.float32)
    lr, gr = run("cpu", torch.float64)
    assert abs(lg - lr) < 1e-5 * abs(lr)
    for k in names:
        torch.testing.assert_close(gg[k], gr[k], rtol=1e-4, atol=1e-6 * float(gr[k].abs().max()) + 1e-9, msg=k)


def test_full_size_config3_one_training_step(oracle):
    """BASELINE configs[2] at full size under pytest: B = 32, 192 x 192 RGB + N = 1024 points, one training step on the GPU
    (end points finite, every trainable parameter receives a finite gradient), and the same model at a B = 4 sub-batch against
    the CPU-oracle composition of the same modules (train-mode BatchNorm couples a batch, so the comparison is run at the
    sub-batch's own statistics): end points and SupervisedLoss within 1e-4."""
    import bench
    from istnet_amd.losses import SupervisedLoss
    from istnet_amd.pointnet2 import pointnet2_utils
    net = bench.make_istnet(DEV, seed=0)
    for m in net.modules():                      # dropout masks come from the device's random stream: off for the comparison
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
            m.eval()
    batch = bench.istnet_batch(32, 1024, seed=0, device=DEV)
    labels = {k: batch[k] for k in ("rotation_label", "translation_label", "size_label", "qo")}
    crit = SupervisedLoss(1.0, 10.0)
    ep = net(batch)
    loss = crit({**ep, **labels})
    assert all(bool(torch.isfinite(v).all()) for v in ep.values()) and bool(torch.isfinite(loss))
    loss.backward()
    bad = [n for n, p in net.named_parameters() if ".fc." not in n and (p.grad is None or not bool(torch.isfinite(p.grad).all()))]
    assert not bad, bad
    # ---- B = 4 sub-batch: GPU vs the CPU composition over the oracle ops, same weights ----
    import copy
    net.zero_grad(set_to_none=True)
    cpu_net = copy.deepcopy(net).to("cpu").to(memory_format=torch.contiguous_format).train()
    for m in cpu_net.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
            m.eval()
    sub = {k: v[:4].contiguous() for k, v in batch.items()}
    sub["rgb"] = sub["rgb"].contiguous(memory_format=torch.channels_last)
    ep_g = net(sub)
    loss_g = crit({**ep_g, **{k: sub[k] for k in labels}})
    saved = pointnet2_utils._ext
    try:
        pointnet2_utils._ext = oracle
        sub_c = {k: v.cpu().contiguous() for k, v in sub.items()}
        ep_c = cpu_net(sub_c)
        loss_c = crit({**ep_c, **{k: sub_c[k] for k in labels}})
    finally:
        pointnet2_utils._ext = saved
    assert set(ep_g) == set(ep_c)
    for k in ep_c:
        want = ep_c[k].detach()
        torch.testing.assert_close(ep_g[k].detach().cpu(), want, rtol=1e-4, atol=1e-4 * float(want.abs().max()), msg=k)
    np.testing.assert_allclose(float(loss_g.detach()), float(loss_c.detach()), rtol=1e-4)


def test_config5_full_batch_indices_and_pose_slice(oracle, ext):
    """BASELINE configs[4] at full size: eval mode, B = 64 instances of N = 2048 points.  (i) every index tensor of the
    encoder's geometry (4 FPS levels, 8 ball queries, 4 three_nn) bit-exact against the oracle for all 64 clouds; (ii) the
    poses of a 4-instance slice of the B = 64 GPU run against the CPU-oracle composition of those 4 instances alone (eval
    mode: no coupling across the batch) within 1e-4."""
    from istnet_amd.ist_net import IST_Net, CAM_RADII
    from istnet_amd.pointnet2 import pointnet2_utils
    b, n = 64, 2048
    pts = _shell(b, n, 21)
    cur_c, cur_g, tie = pts, pts.to(DEV), None
    levels_c, levels_g = [pts], [pts.to(DEV)]
    for li, m in enumerate((512, 256, 128, 64)):
        idx_c = oracle.furthest_point_sampling(cur_c, m)
        nxt = (512, 256, 128, 64)[li + 1] if li < 3 else 0
        idx_g, new_g, tie = ext.furthest_point_sampling_chain(cur_g, m, tie_in=tie, track_rounds=nxt)
        assert torch.equal(idx_g.cpu(), idx_c), f"FPS level {li}"
        new_c = torch.gather(cur_c, 1, idx_c.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        assert torch.equal(new_g.cpu(), new_c)
        for si, (radius, ns) in enumerate(zip(CAM_RADII[li], (16, 32))):
            got = ext.ball_query(new_g, cur_g, radius, ns).cpu()
            assert torch.equal(got, oracle.ball_query(new_c, cur_c, radius, ns)), f"ball query level {li} scale {si}"
        levels_c.append(new_c); levels_g.append(new_g)
        cur_c, cur_g = new_c, new_g
    for lvl in range(3, -1, -1):
        d_g, i_g = ext.three_nn(levels_g[lvl], levels_g[lvl + 1])
        d_c, i_c = oracle.three_nn(levels_c[lvl], levels_c[lvl + 1])
        assert torch.equal(i_g.cpu(), i_c) and torch.equal(d_g.cpu(), d_c), f"three_nn level {lvl}"
    # ---- poses ----
    torch.manual_seed(31)
    net = IST_Net().eval()
    g = torch.Generator().manual_seed(32)
    inputs = {"pts": pts + torch.tensor([0.0, 0.0, 0.8]), "rgb_local": torch.randn(b, 128, n, generator=g),
              "category_label": torch.randint(0, 6, (b, 1), generator=g)}
    net_gpu = IST_Net().to(DEV).eval()
    net_gpu.load_state_dict(net.state_dict())
    with torch.no_grad():
        out_gpu = net_gpu({k: v.to(DEV) for k, v in inputs.items()})
        saved = pointnet2_utils._ext
        try:
            pointnet2_utils._ext = oracle
            sl = slice(17, 21)
            out_cpu = net({k: v[sl].contiguous() for k, v in inputs.items()})
        finally:
            pointnet2_utils._ext = saved
    for k in ("pred_rotation", "pred_translation", "pred_size", "pred_qo"):
        torch.testing.assert_close(out_gpu[k][sl].cpu(), out_cpu[k], **TOL)
