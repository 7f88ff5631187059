"""Host logic (operator wrappers, SA / FP / encoder / heads wiring, state-dict keys) on CPU:
our modules driven over the CPU oracle must reproduce the golden vectors that the REFERENCE
Python produced (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CAM = [[0.01, 0.02], [0.02, 0.04], [0.04, 0.08], [0.08, 0.16]]


def _checksum(sd):
    sd = {k: v for k, v in sd.items() if "running_" not in k and "num_batches" not in k}
    return np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in sd.values()])


def test_query_and_group_config1(cpu_ops):
    z = np.load(os.path.join(GOLD, "config1_sa_grouping.npz"))
    xyz, new_xyz = torch.from_numpy(z["xyz"]), torch.from_numpy(z["new_xyz"])
    fps = cpu_ops.furthest_point_sample(xyz, 512)
    assert np.array_equal(fps.numpy(), z["fps_idx"].astype(np.int32))
    got_new = cpu_ops.gather_operation(xyz.transpose(1, 2).contiguous(), fps).transpose(1, 2).contiguous()
    assert torch.equal(got_new, new_xyz)
    grouped = cpu_ops.QueryAndGroup(0.2, 32)(xyz, new_xyz, None)
    assert grouped.shape == (4, 3, 512, 32)
    assert np.array_equal(grouped.numpy()[:, :, ::16], z["grouped_xyz"])
    assert abs(float(grouped.double().abs().sum()) - float(z["grouped_abs_sum"])) < 1e-6


def test_sa_fp_layer_matches_reference(cpu_ops):
    from istnet_amd.pointnet2.pointnet2_modules import PointnetFPModule, PointnetSAModuleMSG
    z = np.load(os.path.join(GOLD, "sa_fp_layer.npz"))
    torch.manual_seed(20)
    sa = PointnetSAModuleMSG(npoint=64, radii=[0.15, 0.3], nsamples=[8, 16], mlps=[[16, 16, 32], [16, 16, 32]])
    fp = PointnetFPModule(mlp=[64 + 16, 32, 32])
    np.testing.assert_allclose(_checksum({**sa.state_dict(), **fp.state_dict()}), z["state_checksum"], rtol=0, atol=0)
    xyz = torch.from_numpy(z["xyz"])
    feat = torch.from_numpy(z["feat"]).requires_grad_(True)
    new_xyz, new_feat = sa(xyz, feat)
    out = fp(xyz, new_xyz, feat, new_feat)
    out.square().mean().backward()
    assert np.array_equal(new_xyz.detach().numpy(), z["new_xyz"])
    np.testing.assert_allclose(new_feat.detach().numpy(), z["sa_out"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out.detach().numpy(), z["fp_out"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(feat.grad.numpy(), z["grad_feat"], rtol=1e-4, atol=1e-7)
    for name, p in list(sa.named_parameters()) + list(fp.named_parameters()):
        np.testing.assert_allclose(p.grad.numpy(), z["gradp_" + name], rtol=1e-4, atol=1e-7, err_msg=name)


def test_encoder_matches_reference(cpu_ops):
    from istnet_amd.modules import PointNet2MSG
    z = np.load(os.path.join(GOLD, "encoder_b2.npz"))
    torch.manual_seed(0)
    enc = PointNet2MSG([list(r) for r in CAM])
    sd = {k: v for k, v in enc.state_dict().items() if "running" not in k and "num_batches" not in k}
    np.testing.assert_allclose(_checksum(sd), z["state_checksum"], rtol=0, atol=0)
    assert len(enc.state_dict()) == 192 and sum(p.numel() for p in enc.parameters()) == 1307808
    pts = torch.from_numpy(z["pts"])
    enc.train()
    out = enc(pts)
    assert out.shape == (2, 128, 1024)
    np.testing.assert_allclose(out.detach().numpy()[:, :, ::8], z["out_train"], rtol=1e-4, atol=1e-5)
    out.square().mean().backward()
    norms = np.array([float(p.grad.double().norm()) for _, p in enc.named_parameters()])
    np.testing.assert_allclose(norms, z["grad_norms"], rtol=1e-3, atol=1e-9)
    g = dict(enc.named_parameters())
    np.testing.assert_allclose(g["SA_modules.0.mlps.0.layer0.conv.weight"].grad.numpy(), z["grad_first_conv"],
                               rtol=1e-3, atol=1e-8)
    sdn = enc.state_dict()
    np.testing.assert_allclose(sdn["SA_modules.3.mlps.1.layer2.normlayer.bn.running_var"].numpy(),
                               z["running_var_sa3"], rtol=1e-5, atol=1e-8)
    enc.eval()
    with torch.no_grad():
        out_eval = enc(pts)
    np.testing.assert_allclose(out_eval.numpy()[:, :, ::8], z["out_eval"], rtol=1e-4, atol=1e-5)


def test_heads_match_reference():
    from istnet_amd import ist_net, rotation_utils
    z = np.load(os.path.join(GOLD, "ist_heads_b2.npz"))
    t = lambda k: torch.from_numpy(z[k])
    index = t("cls").long() + torch.arange(2) * 6
    cases = {"deformer": (ist_net.FeatureDeformer, (t("pts"), t("rgb_local"), t("pts_local"), index)),
             "light": (ist_net.LightEstimator, (t("pts"), t("rgb_local"), t("pts_local"))),
             "heavy": (ist_net.HeavyEstimator, (t("pts"), t("pts_w"), t("rgb_local"), t("pts_local"), t("pts_w_local")))}
    for name, (ctor, args) in cases.items():
        torch.manual_seed(40)
        m = ctor()
        np.testing.assert_allclose(_checksum(m.state_dict()), z[f"{name}_state_checksum"], rtol=0, atol=0)
        for i, o in enumerate(m(*args)):
            want = z[f"{name}_out{i}"]
            got = o.detach().numpy() if o.numel() < 4096 else o.detach().numpy().reshape(o.shape[0], -1)[:, ::16]
            np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5, err_msg=f"{name}[{i}]")
    rot = rotation_utils.Ortho6d2Mat(t("x6"), t("y6"))
    np.testing.assert_allclose(rot.numpy(), z["rot"], rtol=1e-5, atol=1e-6)
    eye = torch.matmul(rot.transpose(1, 2), rot)
    torch.testing.assert_close(eye, torch.eye(3).expand(5, 3, 3), rtol=1e-5, atol=1e-5)


def test_istnet_point_branch_matches_reference(cpu_ops):
    from istnet_amd.ist_net import IST_Net
    z = np.load(os.path.join(GOLD, "istnet_point_branch_b2.npz"))
    torch.manual_seed(5)
    net = IST_Net()
    np.testing.assert_allclose(_checksum(net.state_dict()), z["state_checksum"], rtol=0, atol=0)
    b, n = 2, 1024
    rgb_feat = torch.from_numpy(z["rgb_feat"])
    choose = torch.from_numpy(z["choose"].astype(np.int64))
    inputs = {"rgb": rgb_feat, "pts": torch.from_numpy(z["pts"]), "choose": choose,
              "category_label": torch.from_numpy(z["cls"]).reshape(b, 1), "qo": torch.from_numpy(z["qo"])}
    net.rgb_cam_extractor = torch.nn.Identity()
    net.train()
    ep = net(inputs)
    sub = lambda v: v.detach().numpy() if v.numel() <= 8192 else v.detach().numpy().reshape(b, -1)[:, ::64]
    keys = [k[len("train_"):] for k in z.files if k.startswith("train_")]
    assert set(keys) == set(ep.keys())
    for k in keys:
        np.testing.assert_allclose(sub(ep[k]), z["train_" + k], rtol=1e-3, atol=1e-5, err_msg=k)
    net.eval()
    with torch.no_grad():
        ev = net(inputs)
    assert set(ev.keys()) == {"pred_qo", "pred_rotation", "pred_translation", "pred_size"}
    for k in ev:
        np.testing.assert_allclose(sub(ev[k]), z["eval_" + k], rtol=1e-4, atol=1e-5, err_msg=k)


def test_rgb_branch_and_loss_match_reference():
    """the RGB branch restated with the reference's module tree, and the training loss."""
    from istnet_amd import losses, rgb_branch
    z = np.load(os.path.join(GOLD, "rgb_branch_and_loss.npz"))
    # the reference consumed the random stream of TWO trunks before the decoder (resnet18() + the weight
    # "download"): build a throw-away trunk first so the streams line up
    torch.manual_seed(60)
    rgb_branch.ResNet()
    net = rgb_branch.ModifiedResnet().eval()
    keys = list(net.state_dict().keys())
    assert len(keys) == int(z["n_keys"]) and keys[:3] + keys[-3:] == list(z["key_first"])
    assert sum(p.numel() for p in net.parameters()) == int(z["n_params"])
    with torch.no_grad():
        out = net(torch.from_numpy(z["img"]))
    assert out.shape == (1, 128, 96, 96)
    np.testing.assert_allclose(out.numpy()[:, ::4, ::6, ::6], z["out_sub"], rtol=1e-4, atol=1e-5)
    ep = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("ep_")}
    # loss terms only use the small tensors; the sub-sampled feature maps keep MSE terms out of this check
    t = lambda k: torch.from_numpy(z["lab_" + k])
    got = losses.PoseDis(ep["pred_rotation"], ep["pred_translation"], ep["pred_size"], t("rotation_label"),
                         t("translation_label"), t("size_label"))
    r = ep["pred_rotation"] - t("rotation_label")
    want = (r.norm(dim=1).mean() + (ep["pred_translation"] - t("translation_label")).norm(dim=1).mean()
            + (ep["pred_size"] - t("size_label")).norm(dim=1).mean())
    torch.testing.assert_close(got, want)
    p1 = torch.tensor([[[0.0, 0.05, 0.3]]]); p2 = torch.zeros(1, 1, 3)
    torch.testing.assert_close(losses.SmoothL1Dis(p1, p2), torch.tensor(0.05 ** 2 / 0.2 + (0.3 - 0.05)))


def test_rgb_local_gather_is_layout_agnostic():
    """IST_Net._rgb_local picks the same per-point RGB features from an NCHW and a channels-last extractor."""
    import torch
    from istnet_amd.ist_net import IST_Net

    class Ext(torch.nn.Module):
        def __init__(self, cl):
            super().__init__()
            self.cl = cl

        def forward(self, rgb):
            f = torch.arange(2 * 128 * 6 * 5, dtype=torch.float32).reshape(2, 128, 6, 5)
            return f.contiguous(memory_format=torch.channels_last) if self.cl else f

    choose = torch.tensor([[0, 7, 29, 3], [29, 1, 1, 15]])
    a = IST_Net(rgb_extractor=Ext(False))._rgb_local({"rgb": None, "choose": choose}, 2)
    b = IST_Net(rgb_extractor=Ext(True))._rgb_local({"rgb": None, "choose": choose}, 2)
    assert a.shape == (2, 128, 4) and torch.equal(a, b)
    assert torch.equal(a[1, :, 3], torch.arange(2 * 128 * 30, dtype=torch.float32).reshape(2, 128, 30)[1, :, 15])


def test_rgb_gather_first_equals_dense_tail_in_eval():
    """Eval mode: `final` applied to the chosen pixels only == the dense (B,128,H,W) map gathered afterwards
    (reference order, ist_net.py:41-45), for NCHW and channels-last; training mode keeps the dense path."""
    import torch
    from istnet_amd import rgb_branch
    from istnet_amd.ist_net import IST_Net
    torch.manual_seed(3)
    ext = rgb_branch.ModifiedResnet()
    for bn in (ext.model.final[1], ext.model.up_3.conv[2]):
        bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 2.0); bn.weight.data.normal_(1, 0.2); bn.bias.data.normal_(0, 0.2)
    ext.eval()
    net = IST_Net(rgb_extractor=ext).eval()
    rgb = torch.randn(2, 3, 48, 48)
    choose = torch.randint(0, 48 * 48, (2, 37))
    choose[0, :6] = torch.tensor([0, 47, 47 * 48, 48 * 48 - 1, 5, 48 * 20])    # corners and edges: the conv's zero padding
    with torch.no_grad():
        dense = ext(rgb)
        want = torch.gather(dense.reshape(2, 128, -1), 2, choose.unsqueeze(1).expand(-1, 128, -1))
        got = net._rgb_local({"rgb": rgb, "choose": choose}, 2)
        ext_cl = ext.to(memory_format=torch.channels_last)
        got_cl = net._rgb_local({"rgb": rgb.contiguous(memory_format=torch.channels_last), "choose": choose}, 2)
    assert got.shape == (2, 128, 37) and got.is_contiguous()
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(got_cl, want, rtol=1e-5, atol=1e-5)
    net.train()
    # training mode: the chosen pixels of the dense map (same dropout masks), in the reference's gather order
    torch.manual_seed(3)
    picked = ext_cl(rgb, choose)
    torch.manual_seed(3)
    dense_train = ext_cl(rgb)
    assert picked.shape == (2, 128, 37)
    torch.testing.assert_close(picked, torch.gather(dense_train.reshape(2, 128, -1), 2, choose.unsqueeze(1).expand(-1, 128, -1)),
                               rtol=1e-5, atol=1e-5)


def _freeze_case():
    from istnet_amd.ist_net import IST_Net
    z = np.load(os.path.join(GOLD, "istnet_freeze_b2.npz"))
    torch.manual_seed(7)
    net = IST_Net(freeze_world_enhancer=True)
    net.rgb_cam_extractor = torch.nn.Identity()
    b = 2
    inputs = {"rgb": torch.from_numpy(z["rgb_feat"]), "pts": torch.from_numpy(z["pts"]),
              "choose": torch.from_numpy(z["choose"].astype(np.int64)),
              "category_label": torch.from_numpy(z["cls"]).reshape(b, 1), "qo": torch.from_numpy(z["qo"])}
    labels = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("lab_")}
    return z, net, inputs, labels


def test_istnet_frozen_world_enhancer_matches_reference(cpu_ops):
    """IST_Net(freeze_world_enhancer=True) -- the second training stage of train.py:102-118: same state-dict keys
    (no world pose head), same train-mode end points (no *_aux_world), same SupervisedLoss value as the reference."""
    from istnet_amd import losses
    z, net, inputs, labels = _freeze_case()
    assert list(net.state_dict().keys()) == [str(k) for k in z["state_keys"]]
    assert not any(k.startswith("world_enhancer.pose_estimator") for k in net.state_dict())
    np.testing.assert_allclose(_checksum(net.state_dict()), z["state_checksum"], rtol=0, atol=0)
    net.train()
    ep = net(inputs)
    keys = [k[len("train_"):] for k in z.files if k.startswith("train_")]
    assert set(keys) == set(ep.keys()) and not any("aux_world" in k for k in ep)
    sub = lambda v: v.detach().numpy() if v.numel() <= 8192 else v.detach().numpy().reshape(2, -1)[:, ::64]
    for k in keys:
        np.testing.assert_allclose(sub(ep[k]), z["train_" + k], rtol=1e-3, atol=1e-5, err_msg=k)
    loss = losses.SupervisedLoss(1.0, 10.0, freeze_world_enhancer=True)({**ep, **labels, "qo": inputs["qo"]})
    np.testing.assert_allclose(float(loss.detach()), float(z["loss"]), rtol=1e-5)


def test_small_host_helpers():
    """group_model_params / RandomDropout / ChamferDis of the reference's utility modules."""
    from istnet_amd import losses
    g = torch.Generator().manual_seed(3)
    a, b = torch.randn(2, 7, 3, generator=g), torch.randn(2, 5, 3, generator=g)
    d = (a.unsqueeze(2) - b.unsqueeze(1)).norm(dim=3)
    want = (0.5 * d.min(2)[0].mean(1) + 0.5 * d.min(1)[0].mean(1)).mean()
    torch.testing.assert_close(losses.ChamferDis(a, b), want, rtol=1e-5, atol=1e-6)
    from istnet_amd.pointnet2 import pointnet2_utils, pytorch_utils
    mlp = pytorch_utils.SharedMLP([3, 8, 8], bn=True)
    groups = pytorch_utils.group_model_params(mlp, lr=0.1, weight_decay=0.01)
    assert [len(g["params"]) for g in groups] == [2, 4] and groups[0]["weight_decay"] == 0.01
    assert groups[1]["weight_decay"] == 0.0 and groups[1]["lr"] == 0.1
    torch.optim.SGD(groups)          # accepted as is
    drop = pointnet2_utils.RandomDropout(p=0.9)
    x = torch.ones(4, 16, 32)
    torch.manual_seed(0)
    y = drop(x)
    assert set(y.unique().tolist()) <= {0.0, 1.0} and 0 < float(y.mean()) <= 1.0     # kept values are not rescaled
    assert torch.equal(drop.eval()(x), x)


def test_pyramid_matrices_restate_adaptive_pool_and_bilinear_upsample():
    """The fixed pooling / upsampling matrices of the GPU form of PSPModule against the framework's own operators."""
    import torch.nn.functional as F
    from istnet_amd import rgb_branch
    sizes, h, w = (1, 2, 3, 6), 24, 20
    pmat, umat, rows = rgb_branch._psp_matrices(sizes, h, w, "cpu")
    x = torch.randn(2, 5, h, w, generator=torch.Generator().manual_seed(0))
    xv = x.permute(0, 2, 3, 1).reshape(2, h * w, 5)
    for s, (r0, r1) in zip(sizes, rows):
        want = F.adaptive_avg_pool2d(x, (s, s)).permute(0, 2, 3, 1).reshape(2, s * s, 5)
        torch.testing.assert_close(torch.matmul(pmat[r0:r1], xv), want, rtol=1e-5, atol=1e-6)
        z = torch.randn(2, 5, s, s, generator=torch.Generator().manual_seed(s))
        want = F.interpolate(z, size=(h, w), mode="bilinear", align_corners=False).permute(0, 2, 3, 1).reshape(2, h * w, 5)
        torch.testing.assert_close(torch.matmul(umat[:, r0:r1], z.permute(0, 2, 3, 1).reshape(2, s * s, 5)), want,
                                   rtol=1e-5, atol=1e-6)


def test_final_stage_at_chosen_pixels_is_exact_in_float64():
    """rgb_branch._FinalAtChosenFn (training-mode `final` at the chosen pixels, BatchNorm batch statistics from the input
    moments, affine backward through the statistics) against the dense composition + gather, in float64 on the host:
    equal to round-off, running statistics included."""
    from istnet_amd import rgb_branch
    torch.manual_seed(0)
    b, c, h, w, n, co = 2, 8, 12, 10, 30, 16
    final = torch.nn.Sequential(torch.nn.Conv2d(c, co, 1), torch.nn.BatchNorm2d(co), torch.nn.PReLU()).double()
    final[1].weight.data.uniform_(0.5, 1.5)
    final[1].bias.data.normal_()
    u = torch.randn(b, c, h, w, dtype=torch.float64).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    choose = torch.randint(0, h * w, (b, n))
    choose[0, :3] = 5
    out = final(u)
    ref = torch.gather(out.reshape(b, co, -1), 2, choose.unsqueeze(1).expand(-1, co, -1))
    dy = torch.randn_like(ref)
    ref.backward(dy)
    want = [u.grad.clone()] + [p.grad.clone() for p in final.parameters()]
    rm, rv = final[1].running_mean.clone(), final[1].running_var.clone()
    u.grad = None
    for p in final.parameters():
        p.grad = None
    final[1].running_mean.zero_()
    final[1].running_var.fill_(1)
    y = rgb_branch._FinalAtChosenFn.apply(u, choose, final[0].weight, final[0].bias, final[1].weight, final[1].bias,
                                          final[2].weight, final[1].running_mean, final[1].running_var, 0.1, 1e-5)
    torch.testing.assert_close(y, ref, rtol=1e-12, atol=1e-12)
    y.backward(dy)
    for a, r in zip([u.grad] + [p.grad for p in final.parameters()], want):
        torch.testing.assert_close(a, r, rtol=1e-10, atol=1e-12)
    torch.testing.assert_close(final[1].running_mean, rm, rtol=1e-12, atol=1e-14)
    torch.testing.assert_close(final[1].running_var, rv, rtol=1e-12, atol=1e-14)


def test_pyramid_module_linear_form_is_exact_in_float64():
    """rgb_branch.PSPModule._forward_linear (bottleneck slices before the upsample, pooling / upsampling as matrix
    products) against the reference composition (modules.py:10-34) in float64 on the host: output and gradients."""
    from istnet_amd import rgb_branch
    torch.manual_seed(1)
    mod = rgb_branch.PSPModule(12, 20).double()
    x = torch.randn(2, 12, 24, 18, dtype=torch.float64).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(2, 20, 24, 18, dtype=torch.float64)
    xr = x.clone(memory_format=torch.preserve_format).requires_grad_(True)
    ref = mod(xr)                                                           # CPU: the reference composition
    ref.backward(dy)
    want = [xr.grad.clone()] + [p.grad.clone() for p in mod.parameters()]
    mod.zero_grad(set_to_none=True)
    xa = x.clone(memory_format=torch.preserve_format).requires_grad_(True)
    got = mod._forward_linear(xa)
    torch.testing.assert_close(got, ref, rtol=1e-6, atol=1e-6)             # the fixed matrices are float32
    got.backward(dy)
    for a, r in zip([xa.grad] + [p.grad for p in mod.parameters()], want):
        torch.testing.assert_close(a, r, rtol=1e-5, atol=1e-5 * float(r.abs().max()))


def test_upsample_then_conv3x3_equals_small_map_mixing_plus_shifted_interpolation():
    """The identity behind rgb_branch.PSPUpsample's GPU form, in float64 on the host:
    conv3x3(U p) = sum_taps shift_tap(U (W_tap p)) with zero padding at the full-size border (what
    istnet_upconv3_fwd_nhwc evaluates from q = p . Wr)."""
    import torch.nn.functional as F
    torch.manual_seed(2)
    b, cin, cout, h, w = 2, 5, 4, 6, 7
    p = torch.randn(b, cin, h, w, dtype=torch.float64)
    weight = torch.randn(cout, cin, 3, 3, dtype=torch.float64)
    bias = torch.randn(cout, dtype=torch.float64)
    ref = F.conv2d(F.interpolate(p, scale_factor=2, mode="bilinear", align_corners=True), weight, bias, padding=1)
    wr = weight.permute(1, 2, 3, 0).reshape(cin, 9 * cout)                       # Wr[ci][(ky*3+kx)*Cout + co]
    q = torch.matmul(p.permute(0, 2, 3, 1).reshape(b * h * w, cin), wr).view(b, h, w, 9, cout)
    out = bias.view(1, cout, 1, 1).expand(b, cout, 2 * h, 2 * w).clone()
    for ky in range(3):
        for kx in range(3):
            up = F.interpolate(q[:, :, :, ky * 3 + kx].permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True)
            up = F.pad(up, (1, 1, 1, 1))                                         # zeros outside the full-size map
            out += up[:, :, ky:ky + 2 * h, kx:kx + 2 * w]                        # position (y + ky - 1, x + kx - 1)
    torch.testing.assert_close(out, ref, rtol=1e-12, atol=1e-12)
