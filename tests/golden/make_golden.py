"""Generate the golden vectors in tests/golden/*.npz by running the REFERENCE's Python modules.

Runs only in the build container (needs /root/reference; never on the GPU box).  The reference's
native ops have no CPU implementation ("CPU not supported"), so its Python composition
(pointnet2_utils / pointnet2_modules / pytorch_utils / modules.PointNet2MSG / ist_net heads /
rotation_utils) is imported unmodified and driven over ``oracle/pn2_oracle`` standing in for
``pointnet2._ext`` -- three harness-side shims, no edits to the reference:
  1. sys.modules['pointnet2'] / ['pointnet2._ext'] pre-registered (package path model/pointnet2,
     _ext = the CPU oracle);
  2. ``Tensor.cuda`` / ``Module.cuda`` made no-ops (ist_net.py:38, rotation_utils.py:6);
  3. sys.path as in train.py:10-14.
What this pins: the COMPOSITION (grouping order, concat order, BN/ReLU placement, weight formula,
head wiring, state-dict keys).  It does not pin kernel semantics (see oracle/pn2_oracle.c header).

Only data is stored: seeded inputs, index tensors, outputs / slices / checksums.

    python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import pn2_oracle  # noqa: E402


def import_reference():
    for sub in ("provider", "model", os.path.join("model", "pointnet2"), "utils"):
        sys.path.insert(0, os.path.join(REF, sub))
    pkg = types.ModuleType("pointnet2")
    pkg.__path__ = [os.path.join(REF, "model", "pointnet2")]
    sys.modules["pointnet2"] = pkg
    sys.modules["pointnet2._ext"] = pn2_oracle
    pkg._ext = pn2_oracle
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    import pointnet2.pointnet2_utils as ref_utils
    import pointnet2.pointnet2_modules as ref_modules
    import modules as ref_model_modules
    import ist_net as ref_ist
    import rotation_utils as ref_rot
    return ref_utils, ref_modules, ref_model_modules, ref_ist, ref_rot


def npy(t):
    return t.detach().cpu().numpy()


class F64Ops:
    """``pointnet2._ext`` stand-in for a float64 run of the reference encoder: index decisions come from the fp32
    oracle (so they are the ones every fp32 implementation takes), every feature-valued op is torch float64.
    Forward only.  Gives the arithmetic 'truth' the fp32 outputs are within rounding of."""

    def __init__(self, ops):
        self.ops = ops

    def furthest_point_sampling(self, xyz, m):
        return self.ops.furthest_point_sampling(xyz.float().contiguous(), m)

    def ball_query(self, new_xyz, xyz, radius, nsample):
        return self.ops.ball_query(new_xyz.float().contiguous(), xyz.float().contiguous(), radius, nsample)

    def three_nn(self, unknown, known):
        d2, idx = self.ops.three_nn(unknown.float().contiguous(), known.float().contiguous())
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


def state_checksum(sd):
    """Order-sensitive fingerprint of a state dict: per-tensor (sum, abs-sum) in float64."""
    return np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in sd.values()])


def params_only(sd):
    """BatchNorm running statistics change during the forward passes below: fingerprint parameters only."""
    return {k: v for k, v in sd.items() if "running_" not in k and "num_batches" not in k}


def shell_cloud(b, n, seed):
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(b, n, 3, generator=g)
    pts = d / d.norm(dim=2, keepdim=True) * 0.1 + torch.randn(b, n, 3, generator=g) * 0.002
    return (pts - pts.mean(dim=1, keepdim=True)).contiguous()


def main():
    ref_utils, ref_modules, ref_model_modules, ref_ist, ref_rot = import_reference()
    import istnet_amd  # noqa: F401
    from istnet_amd.pointnet2 import pointnet2_utils as my_utils
    my_utils._ext = pn2_oracle  # generation-time cross-check of our host logic on CPU
    from istnet_amd.pointnet2 import pointnet2_modules as my_modules
    from istnet_amd import modules as my_model_modules, ist_net as my_ist, rotation_utils as my_rot

    # ---- 1. BASELINE config 1: single SA grouping on B=4 N=1024 ---------------------------
    xyz = torch.rand(4, 1024, 3, generator=torch.Generator().manual_seed(0))
    fps = ref_utils.furthest_point_sample(xyz, 512)
    new_xyz = ref_utils.gather_operation(xyz.transpose(1, 2).contiguous(), fps).transpose(1, 2).contiguous()
    bq = ref_utils.ball_query(0.2, 32, xyz, new_xyz)
    grouped = ref_utils.QueryAndGroup(0.2, 32)(xyz, new_xyz, None)
    assert torch.equal(my_utils.QueryAndGroup(0.2, 32)(xyz, new_xyz, None), grouped)
    np.savez_compressed(os.path.join(HERE, "config1_sa_grouping.npz"),
                        xyz=npy(xyz), fps_idx=npy(fps).astype(np.int16), new_xyz=npy(new_xyz),
                        ball_idx=npy(bq).astype(np.int16), grouped_xyz=npy(grouped).astype(np.float32)[:, :, ::16],
                        grouped_sum=np.float64(grouped.double().sum()),
                        grouped_abs_sum=np.float64(grouped.double().abs().sum()))

    # ---- 2. one SA-MSG layer and one FP layer, fwd + grads --------------------------------
    g = torch.Generator().manual_seed(2)
    xyz2 = torch.rand(2, 256, 3, generator=g)
    feat2 = torch.randn(2, 16, 256, generator=g)
    torch.manual_seed(20)
    sa_ref = ref_modules.PointnetSAModuleMSG(npoint=64, radii=[0.15, 0.3], nsamples=[8, 16],
                                             mlps=[[16, 16, 32], [16, 16, 32]])
    fp_ref = ref_modules.PointnetFPModule(mlp=[64 + 16, 32, 32])
    torch.manual_seed(20)
    sa_my = my_modules.PointnetSAModuleMSG(npoint=64, radii=[0.15, 0.3], nsamples=[8, 16],
                                           mlps=[[16, 16, 32], [16, 16, 32]])
    fp_my = my_modules.PointnetFPModule(mlp=[64 + 16, 32, 32])
    for a, b in ((sa_ref, sa_my), (fp_ref, fp_my)):
        assert list(a.state_dict().keys()) == list(b.state_dict().keys())
        for k in a.state_dict():
            assert torch.equal(a.state_dict()[k], b.state_dict()[k]), k

    def run_sa_fp(sa, fp):
        f = feat2.clone().requires_grad_(True)
        nx, nf = sa(xyz2, f)
        out = fp(xyz2, nx, f, nf)
        out.square().mean().backward()
        return nx, nf, out, f.grad, {n: p.grad.clone() for n, p in list(sa.named_parameters()) + list(fp.named_parameters())}

    nx_r, nf_r, out_r, gf_r, gp_r = run_sa_fp(sa_ref, fp_ref)
    nx_m, nf_m, out_m, gf_m, gp_m = run_sa_fp(sa_my, fp_my)
    assert torch.equal(nx_r, nx_m)
    torch.testing.assert_close(out_m, out_r, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(gf_m, gf_r, rtol=1e-5, atol=1e-7)
    np.savez_compressed(os.path.join(HERE, "sa_fp_layer.npz"), xyz=npy(xyz2), feat=npy(feat2),
                        new_xyz=npy(nx_r), sa_out=npy(nf_r), fp_out=npy(out_r), grad_feat=npy(gf_r),
                        **{"gradp_" + k: npy(v) for k, v in gp_r.items()},
                        state_checksum=state_checksum(params_only({**sa_ref.state_dict(), **fp_ref.state_dict()})))

    # ---- 3. PointNet2MSG encoder, B=2 N=1024, train and eval BN ---------------------------
    cam = [[0.01, 0.02], [0.02, 0.04], [0.04, 0.08], [0.08, 0.16]]
    pts = shell_cloud(2, 1024, seed=3)
    torch.manual_seed(0)
    enc_ref = ref_model_modules.PointNet2MSG(radii_list=[list(r) for r in cam])
    torch.manual_seed(0)
    enc_my = my_model_modules.PointNet2MSG(radii_list=[list(r) for r in cam])
    assert list(enc_ref.state_dict().keys()) == list(enc_my.state_dict().keys())
    for k, v in enc_ref.state_dict().items():
        assert torch.equal(v, enc_my.state_dict()[k]), k

    captured = {}
    orig = {n: getattr(pn2_oracle, n) for n in ("furthest_point_sampling", "ball_query", "three_nn")}
    counters = {n: 0 for n in orig}

    def tap(name):
        def fn(*a, **k):
            res = orig[name](*a, **k)
            i = counters[name]
            counters[name] += 1
            if name == "three_nn":
                captured[f"three_nn_dist2_{i}"] = npy(res[0])
                captured[f"three_nn_idx_{i}"] = npy(res[1]).astype(np.int16)
            else:
                captured[f"{name}_{i}"] = npy(res).astype(np.int16)
            return res
        return fn

    # float64 evaluation of the reference encoder (same initial weights, same index decisions)
    import copy
    enc64 = copy.deepcopy(enc_ref).double().train()
    saved_ext = ref_utils._ext
    ref_utils._ext = F64Ops(pn2_oracle)
    try:
        with torch.no_grad():
            out_train_f64 = enc64(pts.double())
    finally:
        ref_utils._ext = saved_ext

    enc_ref.train()
    for n in orig:
        setattr(pn2_oracle, n, tap(n))
    try:
        out_train = enc_ref(pts)
    finally:
        for n, f in orig.items():
            setattr(pn2_oracle, n, f)
    out_train.square().mean().backward()
    grads = {n: p.grad.clone() for n, p in enc_ref.named_parameters()}
    running = {k: v.clone() for k, v in enc_ref.state_dict().items() if "running" in k}
    enc_my.train()
    out_my = enc_my(pts)
    out_my.square().mean().backward()
    torch.testing.assert_close(out_my, out_train, rtol=1e-5, atol=1e-6)
    print("encoder_b2: max |reference fp32 - reference fp64| =",
          float((out_train.double() - out_train_f64).abs().max()))
    for n, p in enc_my.named_parameters():
        torch.testing.assert_close(p.grad, grads[n], rtol=1e-4, atol=1e-7)
    enc_ref.eval()
    with torch.no_grad():
        out_eval = enc_ref(pts)
    np.savez_compressed(
        os.path.join(HERE, "encoder_b2.npz"), pts=npy(pts),
        out_train=npy(out_train)[:, :, ::8], out_train_f64=npy(out_train_f64)[:, :, ::8],
        out_train_sum=np.float64(out_train.double().sum()),
        out_train_abs_sum=np.float64(out_train.double().abs().sum()),
        out_eval=npy(out_eval)[:, :, ::8], out_eval_abs_sum=np.float64(out_eval.double().abs().sum()),
        grad_norms=np.array([float(grads[n].double().norm()) for n, _ in enc_ref.named_parameters()]),
        grad_first_conv=npy(grads["SA_modules.0.mlps.0.layer0.conv.weight"]),
        grad_last_fp_conv=npy(grads["FP_modules.0.mlp.layer1.conv.weight"]),
        running_mean_sa3=npy(running["SA_modules.3.mlps.1.layer2.normlayer.bn.running_mean"]),
        running_var_sa3=npy(running["SA_modules.3.mlps.1.layer2.normlayer.bn.running_var"]),
        state_checksum=state_checksum(params_only(enc_my.state_dict())),
        **captured)

    # ---- 4. IST head / pose heads / Ortho6d2Mat on B=2 -------------------------------------
    g = torch.Generator().manual_seed(4)
    pts4 = shell_cloud(2, 256, seed=4)
    rgb_local = torch.randn(2, 128, 256, generator=g)
    pts_local = torch.randn(2, 128, 256, generator=g)
    pts_w_local = torch.randn(2, 128, 256, generator=g)
    pts_w = torch.rand(2, 256, 3, generator=g) - 0.5
    cls = torch.tensor([1, 4])
    index = cls + torch.arange(2) * 6
    out4 = {}
    for name, ctor_r, ctor_m in (("deformer", ref_ist.FeatureDeformer, my_ist.FeatureDeformer),
                                 ("light", ref_ist.LightEstimator, my_ist.LightEstimator),
                                 ("heavy", ref_ist.HeavyEstimator, my_ist.HeavyEstimator)):
        torch.manual_seed(40)
        m_r = ctor_r()
        torch.manual_seed(40)
        m_m = ctor_m()
        assert list(m_r.state_dict().keys()) == list(m_m.state_dict().keys()), name
        for k, v in m_r.state_dict().items():
            assert torch.equal(v, m_m.state_dict()[k]), (name, k)
        if name == "deformer":
            args = (pts4, rgb_local, pts_local, index)
        elif name == "light":
            args = (pts4, rgb_local, pts_local)
        else:
            args = (pts4, pts_w, rgb_local, pts_local, pts_w_local)
        res_r, res_m = m_r(*args), m_m(*args)
        for i, (a, b) in enumerate(zip(res_r, res_m)):
            torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-6)
            out4[f"{name}_out{i}"] = npy(a) if a.numel() < 4096 else npy(a).reshape(a.shape[0], -1)[:, ::16]
            out4[f"{name}_out{i}_abs_sum"] = np.float64(a.double().abs().sum())
        out4[f"{name}_state_checksum"] = state_checksum(m_r.state_dict())
    x6 = torch.randn(5, 3, generator=g)
    y6 = torch.randn(5, 3, generator=g)
    rot = ref_rot.Ortho6d2Mat(x6, y6)
    torch.testing.assert_close(my_rot.Ortho6d2Mat(x6, y6), rot, rtol=1e-6, atol=1e-7)
    np.savez_compressed(os.path.join(HERE, "ist_heads_b2.npz"), pts=npy(pts4), rgb_local=npy(rgb_local),
                        pts_local=npy(pts_local), pts_w_local=npy(pts_w_local), pts_w=npy(pts_w),
                        cls=npy(cls), x6=npy(x6), y6=npy(y6), rot=npy(rot), **out4)

    # ---- 5. IST_Net point branch end to end (rgb features supplied), train + eval ----------
    torch.manual_seed(5)
    net_r = ref_ist.IST_Net.__new__(ref_ist.IST_Net)
    torch.nn.Module.__init__(net_r)
    net_r.nclass, net_r.freeze_world_enhancer = 6, False
    net_r.rgb_cam_extractor = torch.nn.Identity()  # RGB branch out of scope: feed features as "rgb"
    net_r.pts_cam_extractor = ref_model_modules.PointNet2MSG(radii_list=[[0.01, 0.02], [0.02, 0.04], [0.04, 0.08], [0.08, 0.16]])
    net_r.implicit_transform = ref_ist.ImplicitTransformation(6)
    net_r.main_estimator = ref_ist.HeavyEstimator()
    net_r.cam_enhancer = ref_ist.LightEstimator()
    net_r.world_enhancer = ref_ist.WorldSpaceEnhancer(freeze=False)
    torch.manual_seed(5)
    net_m = my_ist.IST_Net()
    sd_r = {k: v for k, v in net_r.state_dict().items()}
    assert list(sd_r.keys()) == list(net_m.state_dict().keys())
    net_m.load_state_dict(sd_r)
    g = torch.Generator().manual_seed(50)
    b, n = 2, 1024
    pts5 = shell_cloud(b, n, seed=51) + torch.tensor([0.0, 0.0, 0.8])
    rgb_feat = torch.randn(b, 128, 24, 24, generator=g)
    choose = torch.randint(0, 24 * 24, (b, n), generator=g)
    inputs = {"rgb": rgb_feat, "pts": pts5, "choose": choose, "category_label": torch.tensor([[2], [5]]),
              "qo": torch.rand(b, n, 3, generator=g) - 0.5}
    net_r.train(); net_m.train()
    ep_r = net_r(inputs)
    ep_m = net_m({**inputs, "rgb_local": torch.gather(rgb_feat.view(b, 128, -1), 2, choose.unsqueeze(1).repeat(1, 128, 1)).contiguous()})
    assert set(ep_r.keys()) == set(ep_m.keys())
    store = {}
    for k in sorted(ep_r.keys()):
        torch.testing.assert_close(ep_m[k], ep_r[k], rtol=1e-4, atol=1e-5)
        store["train_" + k] = npy(ep_r[k]) if ep_r[k].numel() <= 8192 else npy(ep_r[k]).reshape(b, -1)[:, ::64]
    net_r.eval(); net_m.eval()
    with torch.no_grad():
        ev_r = net_r(inputs)
    for k in sorted(ev_r.keys()):
        store["eval_" + k] = npy(ev_r[k]) if ev_r[k].numel() <= 8192 else npy(ev_r[k]).reshape(b, -1)[:, ::64]
    np.savez_compressed(os.path.join(HERE, "istnet_point_branch_b2.npz"), pts=npy(pts5), rgb_feat=npy(rgb_feat).astype(np.float32),
                        choose=npy(choose).astype(np.int16), cls=np.array([2, 5]), qo=npy(inputs["qo"]),
                        state_checksum=state_checksum(params_only(sd_r)), **store)
    # ---- 6. RGB branch (out of the hot path; module-tree / state-dict parity) and the training loss -----
    import resnet as ref_resnet
    import losses as ref_losses
    from istnet_amd import rgb_branch as my_rgb, losses as my_losses
    ref_resnet.model_zoo.load_url = lambda *a, **k: ref_resnet.ResNet(ref_resnet.BasicBlock, [2, 2, 2, 2]).state_dict()
    torch.manual_seed(60)
    rgb_r = ref_model_modules.ModifiedResnet()
    # the reference builds a SECOND ResNet inside the load_url shim above (RNG advanced, weights replaced):
    # rebuild ours the same way so the random streams line up, then copy the trunk state as load_state_dict did
    torch.manual_seed(60)
    rgb_m = my_rgb.ModifiedResnet()
    assert list(rgb_r.state_dict().keys()) == list(rgb_m.state_dict().keys())
    rgb_m.load_state_dict(rgb_r.state_dict())
    img = torch.randn(1, 3, 96, 96, generator=torch.Generator().manual_seed(61))
    rgb_r.eval(); rgb_m.eval()
    with torch.no_grad():
        o_r, o_m = rgb_r(img), rgb_m(img)
    torch.testing.assert_close(o_m, o_r, rtol=1e-5, atol=1e-6)
    keys = list(rgb_r.state_dict().keys())
    # loss on the train-mode end points of section 5
    class _Cfg: pass
    cfg = _Cfg(); cfg.loss = _Cfg(); cfg.loss.gamma1, cfg.loss.gamma2, cfg.freeze_world_enhancer = 1.0, 10.0, False
    g6 = torch.Generator().manual_seed(62)
    lab = {"rotation_label": torch.linalg.qr(torch.randn(2, 3, 3, generator=g6))[0], "translation_label": torch.randn(2, 3, generator=g6),
           "size_label": torch.rand(2, 3, generator=g6), "qo": inputs["qo"]}
    net_r.train(); net_m.train()
    torch.manual_seed(63)
    ep_r = net_r(inputs); ep_r.update(lab)
    loss_r = ref_ist.SupervisedLoss(cfg)(ep_r)
    ep_chk = {k: v for k, v in ep_r.items()}
    loss_m = my_losses.SupervisedLoss(1.0, 10.0, False)(ep_chk)
    torch.testing.assert_close(loss_m, loss_r, rtol=1e-6, atol=1e-7)
    np.savez_compressed(os.path.join(HERE, "rgb_branch_and_loss.npz"), img=npy(img), out_sub=npy(o_r)[:, ::4, ::6, ::6],
                        out_abs_sum=np.float64(o_r.double().abs().sum()), n_keys=np.int64(len(keys)),
                        key_first=np.array(keys[:3] + keys[-3:]),
                        n_params=np.int64(sum(p.numel() for p in rgb_r.parameters())),
                        loss=np.float64(loss_r.item()),
                        **{"lab_" + k: npy(v) for k, v in lab.items() if k != "qo"},
                        **{"ep_" + k: (npy(v) if v.numel() <= 8192 else npy(v).reshape(2, -1)[:, ::64]) for k, v in ep_r.items() if k not in lab})
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
