"""Golden vectors for the frozen-world-enhancer variant of IST-Net (model/ist_net.py:11-20,63-68,102-110; the
configuration train.py:102-118 trains in its second stage), produced by the REFERENCE's Python modules over the CPU
oracle ops -- same harness as make_golden.py (build container only).  Only data is stored.

    python tests/golden/make_golden_freeze.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    _, _, ref_model_modules, ref_ist, _ = mg.import_reference()
    torch.manual_seed(7)
    net = ref_ist.IST_Net.__new__(ref_ist.IST_Net)
    torch.nn.Module.__init__(net)
    net.nclass, net.freeze_world_enhancer = 6, True
    net.rgb_cam_extractor = torch.nn.Identity()      # RGB features are fed as "rgb" (the branch itself is torch.nn on both sides)
    net.pts_cam_extractor = ref_model_modules.PointNet2MSG(radii_list=[[0.01, 0.02], [0.02, 0.04], [0.04, 0.08], [0.08, 0.16]])
    net.implicit_transform = ref_ist.ImplicitTransformation(6)
    net.main_estimator = ref_ist.HeavyEstimator()
    net.cam_enhancer = ref_ist.LightEstimator()
    net.world_enhancer = ref_ist.WorldSpaceEnhancer(freeze=True)
    g = torch.Generator().manual_seed(70)
    b, n = 2, 512
    pts = mg.shell_cloud(b, n, seed=71) + torch.tensor([0.0, 0.0, 0.8])
    rgb_feat = torch.randn(b, 128, 16, 16, generator=g)
    choose = torch.randint(0, 256, (b, n), generator=g)
    rot = torch.linalg.qr(torch.randn(b, 3, 3, generator=g))[0]
    inputs = {"rgb": rgb_feat, "pts": pts, "choose": choose, "category_label": torch.tensor([[1], [4]]),
              "qo": torch.rand(b, n, 3, generator=g) - 0.5}
    labels = {"rotation_label": rot, "translation_label": pts.mean(dim=1), "size_label": torch.rand(b, 3, generator=g) * 0.2 + 0.05}
    net.train()
    ep = net(inputs)
    cfg = types.SimpleNamespace(loss=types.SimpleNamespace(gamma1=1.0, gamma2=10.0), freeze_world_enhancer=True)
    loss = ref_ist.SupervisedLoss(cfg)({**ep, **labels, "qo": inputs["qo"]})
    store = {"train_" + k: (mg.npy(v) if v.numel() <= 8192 else mg.npy(v).reshape(b, -1)[:, ::64]) for k, v in ep.items()}
    np.savez_compressed(os.path.join(HERE, "istnet_freeze_b2.npz"), pts=mg.npy(pts), rgb_feat=mg.npy(rgb_feat),
                        choose=mg.npy(choose).astype(np.int16), cls=np.array([1, 4]), qo=mg.npy(inputs["qo"]),
                        state_keys=np.array(list(net.state_dict().keys())),
                        state_checksum=mg.state_checksum(mg.params_only(net.state_dict())), loss=mg.npy(loss),
                        **{"lab_" + k: mg.npy(v) for k, v in labels.items()}, **store)
    print(sorted(store), float(loss), len(net.state_dict()))


if __name__ == "__main__":
    main()
