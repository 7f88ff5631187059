"""Float64 evaluations of the REFERENCE's point branch for the two full-model fixtures (istnet_point_branch_b2.npz,
istnet_freeze_b2.npz): the same reference modules, initial weights and inputs as make_golden.py / make_golden_freeze.py,
run in double precision with the index decisions (FPS picks, ball-query and three_nn indices) REPLAYED from the float32
run -- so the float64 result is the arithmetic truth of exactly the composition every float32 implementation evaluates.
The float32 reference itself is up to a few 1e-4 away from it in train mode (printed below): the GPU tests hold the HIP
path to 1e-4 of THIS result (tests/test_golden_gpu.py), which a comparison against another float32 evaluation cannot do.
Build container only; stores data only.

    python tests/golden/make_golden_f64.py
"""
import copy
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from oracle import pn2_oracle  # noqa: E402

INDEX_OPS = ("furthest_point_sampling", "ball_query", "three_nn")


class Recorder:
    """pointnet2._ext stand-in for the float32 run: the oracle, with every index result recorded in call order."""

    def __init__(self):
        self.log = []

    def __getattr__(self, name):
        fn = getattr(pn2_oracle, name)
        if name not in INDEX_OPS:
            return fn

        def wrapped(*a, **k):
            res = fn(*a, **k)
            self.log.append((name, res))
            return res
        return wrapped


class Replay(mg.F64Ops):
    """float64 run: feature ops in torch float64 (F64Ops), index ops answered from the float32 run's log."""

    def __init__(self, log):
        super().__init__(pn2_oracle)
        self.log, self.pos = log, 0

    def _next(self, name):
        got, res = self.log[self.pos]
        assert got == name, (got, name)
        self.pos += 1
        return res

    def furthest_point_sampling(self, xyz, m):
        return self._next("furthest_point_sampling")

    def ball_query(self, new_xyz, xyz, radius, nsample):
        return self._next("ball_query")

    def three_nn(self, unknown, known):
        _, idx = self._next("three_nn")
        picked = torch.gather(known.unsqueeze(1).expand(-1, unknown.size(1), -1, -1), 2,
                              idx.long().unsqueeze(-1).expand(-1, -1, -1, 3))
        return (unknown.unsqueeze(2) - picked).pow(2).sum(-1), idx


def build(ref_model_modules, ref_ist, seed, freeze):
    torch.manual_seed(seed)
    net = ref_ist.IST_Net.__new__(ref_ist.IST_Net)
    torch.nn.Module.__init__(net)
    net.nclass, net.freeze_world_enhancer = 6, freeze
    net.rgb_cam_extractor = torch.nn.Identity()
    net.pts_cam_extractor = ref_model_modules.PointNet2MSG(radii_list=[[0.01, 0.02], [0.02, 0.04], [0.04, 0.08], [0.08, 0.16]])
    net.implicit_transform = ref_ist.ImplicitTransformation(6)
    net.main_estimator = ref_ist.HeavyEstimator()
    net.cam_enhancer = ref_ist.LightEstimator()
    net.world_enhancer = ref_ist.WorldSpaceEnhancer(freeze=freeze)
    return net


def both_precisions(ref_utils, net, inputs, stored, b):
    sub = lambda v: mg.npy(v) if v.numel() <= 8192 else mg.npy(v).reshape(b, -1)[:, ::64]
    net64 = copy.deepcopy(net).double().train()
    rec = Recorder()
    saved = ref_utils._ext
    ref_utils._ext = rec
    try:
        net.train()
        ep32 = net(inputs)
    finally:
        ref_utils._ext = saved
    for k, v in ep32.items():        # the float32 run IS the committed fixture
        np.testing.assert_array_equal(sub(v), stored["train_" + k], err_msg=k)
    replay = Replay(rec.log)
    ref_utils._ext = replay
    try:
        with torch.no_grad():
            ep64 = net64({k: (v.double() if v.is_floating_point() else v) for k, v in inputs.items()})
    finally:
        ref_utils._ext = saved
    assert replay.pos == len(rec.log)
    worst = {k: float((ep32[k].double() - ep64[k]).abs().max() / (ep64[k].abs().max() + 1e-30)) for k in ep64}
    return {k: sub(v) for k, v in ep64.items()}, ep64, worst


def main():
    ref_utils, _, ref_model_modules, ref_ist, _ = mg.import_reference()
    out = {}
    # ---- istnet_point_branch_b2.npz (make_golden.py section 5) ----
    z = np.load(os.path.join(HERE, "istnet_point_branch_b2.npz"))
    net = build(ref_model_modules, ref_ist, seed=5, freeze=False)
    g = torch.Generator().manual_seed(50)
    b, n = 2, 1024
    pts = mg.shell_cloud(b, n, seed=51) + torch.tensor([0.0, 0.0, 0.8])
    rgb_feat = torch.randn(b, 128, 24, 24, generator=g)
    choose = torch.randint(0, 24 * 24, (b, n), generator=g)
    inputs = {"rgb": rgb_feat, "pts": pts, "choose": choose, "category_label": torch.tensor([[2], [5]]),
              "qo": torch.rand(b, n, 3, generator=g) - 0.5}
    store, _, worst = both_precisions(ref_utils, net, inputs, z, b)
    print("point branch, train mode: max |reference fp32 - fp64| / max |fp64| per end point")
    for k, v in sorted(worst.items()):
        print(f"   {k:32s} {v:.2e}")
    out.update({"pb_" + k: v for k, v in store.items()})
    # ---- istnet_freeze_b2.npz (make_golden_freeze.py) ----
    z = np.load(os.path.join(HERE, "istnet_freeze_b2.npz"))
    net = build(ref_model_modules, ref_ist, seed=7, freeze=True)
    g = torch.Generator().manual_seed(70)
    b, n = 2, 512
    pts = mg.shell_cloud(b, n, seed=71) + torch.tensor([0.0, 0.0, 0.8])
    rgb_feat = torch.randn(b, 128, 16, 16, generator=g)
    choose = torch.randint(0, 256, (b, n), generator=g)
    rot = torch.linalg.qr(torch.randn(b, 3, 3, generator=g))[0]
    inputs = {"rgb": rgb_feat, "pts": pts, "choose": choose, "category_label": torch.tensor([[1], [4]]),
              "qo": torch.rand(b, n, 3, generator=g) - 0.5}
    labels = {"rotation_label": rot, "translation_label": pts.mean(dim=1), "size_label": torch.rand(b, 3, generator=g) * 0.2 + 0.05}
    store, ep64, worst = both_precisions(ref_utils, net, inputs, z, b)
    cfg = types.SimpleNamespace(loss=types.SimpleNamespace(gamma1=1.0, gamma2=10.0), freeze_world_enhancer=True)
    loss64 = ref_ist.SupervisedLoss(cfg)({**ep64, **{k: v.double() for k, v in labels.items()}, "qo": inputs["qo"].double()})
    print("frozen world enhancer, train mode:")
    for k, v in sorted(worst.items()):
        print(f"   {k:32s} {v:.2e}")
    print("   loss fp32", float(z["loss"]), "fp64", float(loss64))
    out.update({"fz_" + k: v for k, v in store.items()})
    out["fz_loss"] = mg.npy(loss64)
    np.savez_compressed(os.path.join(HERE, "istnet_point_branch_f64.npz"), **out)
    print("stored", len(out), "arrays")


if __name__ == "__main__":
    main()
