"""Golden vectors for the f2 host pipeline from the REFERENCE's own code (build container only).

Imports /root/reference/provider/data_augmentation.py and /root/reference/utils/data_utils.py unmodified -- both only
need ``cv2`` at import time (neither function below calls it), so an empty stub module named ``cv2`` is enough -- and
stores inputs / outputs of

* ``data_augment`` (data_augmentation.py:217-285, called from provider/dataset.py:277-283) for every branch: the
  bounding-box deformation for symmetric and asymmetric classes, the rigid perturbation, the box-cage resize (mug, bowl),
  the point noise and the non-linear deformation (both axes), alone, all together, and with the branch probabilities of
  config/ist_net_default.yaml under many seeds.  The reference draws its random numbers from torch's global CPU
  generator; this script replays the same draws in the same order (same seed) and stores them as INPUTS, so an
  implementation that takes the draws as arguments can be compared value for value;
* the symmetric-class rotation canonicalisation and the NOCS coordinates ``qo`` (provider/dataset.py:236-245, restated
  here from that block: the surrounding ``__getitem__`` needs cv2 / torchvision / the dataset on disk) are NOT taken
  from the reference and are therefore not part of this fixture;
* ``get_bbox`` (data_utils.py:43-71) on 4 000 random detection boxes, including boxes at and beyond the image border.

Only data is stored.        python tests/golden/make_golden_augment.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


class Args:
    def __init__(self, bb, rt, bc, pc, nl, pc_r=0.002):
        self.aug_bb_pro, self.aug_rt_pro, self.aug_bc_pro, self.aug_pc_pro, self.aug_nl_pro = bb, rt, bc, pc, nl
        self.aug_pc_r = pc_r


def random_rotation(rng):
    q, r = np.linalg.qr(rng.standard_normal((3, 3)))
    q = q * np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q.astype(np.float32)


SYM = {0: [1, 1, 0, 1], 1: [1, 1, 0, 1], 2: [0, 0, 0, 0], 3: [1, 1, 1, 1], 4: [0, 1, 0, 0], 5: [0, 1, 0, 0]}   # dataset.py:135-157


def replay_draws(args, obj_id, n, seed):
    """The numbers data_augment will draw from torch's global generator after manual_seed(seed), in its order."""
    torch.manual_seed(seed)
    prop = torch.zeros(5)
    bc, nl, noise = torch.zeros(2), torch.zeros(2), torch.zeros(n, 3)
    prop[0] = torch.rand(1)
    prop[1] = torch.rand(1)
    prop[2] = torch.rand(1)
    if prop[2] < args.aug_bc_pro and obj_id in (5, 1):
        bc[0], bc[1] = torch.rand(1), torch.rand(1)
    prop[3] = torch.rand(1)
    if prop[3] < args.aug_pc_pro:
        noise = torch.randn(n, 3)
    prop[4] = torch.rand(1)
    if prop[4] < args.aug_nl_pro and obj_id in (0, 1, 2, 3, 5):
        nl[0], nl[1] = torch.rand(1), torch.rand(1)
    return prop, bc, nl, noise


def main():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))          # imported, never called by the functions used here
    sys.path.insert(0, os.path.join(REF, "provider"))
    sys.path.insert(0, os.path.join(REF, "utils"))
    import data_augmentation as ref_aug
    import data_utils as ref_du

    rng = np.random.default_rng(11)
    n, m = 96, 160
    cases = []
    one = dict(bb=0.0, rt=0.0, bc=0.0, pc=0.0, nl=0.0)
    for obj in range(6):
        for name in ("bb", "rt", "bc", "pc", "nl"):
            cases.append((obj, Args(**{**one, name: 1.0})))
        cases.append((obj, Args(1.0, 1.0, 1.0, 1.0, 1.0)))
    for k in range(36):                                            # the shipped configuration: bb 0.3, rt 0.3, rest 0
        cases.append((k % 6, Args(0.3, 0.3, 0.0, 0.0, 0.0)))
    for k in range(24):                                            # every branch with probability one half
        cases.append((k % 6, Args(0.5, 0.5, 0.5, 0.5, 0.5)))

    keys_in = ("pts", "R", "t", "s", "sym", "aug_bb", "aug_rt_t", "aug_rt_r", "model", "qo", "obj_id", "probs", "pc_r",
               "prop", "bc", "nl", "noise", "seed")
    keys_out = ("out_pts", "out_R", "out_t", "out_s", "out_model", "out_qo")
    store = {k: [] for k in keys_in + keys_out}
    for ci, (obj, args) in enumerate(cases):
        R = random_rotation(rng)
        s = rng.uniform(0.05, 0.35, 3).astype(np.float32)
        t = (rng.uniform(-0.3, 0.3, 3) + np.array([0, 0, 0.9])).astype(np.float32)
        model = (rng.uniform(-0.5, 0.5, (m, 3)) * s / np.linalg.norm(s)).astype(np.float32)       # NOCS model points
        qo = (rng.uniform(-0.5, 0.5, (n, 3)) * s / np.linalg.norm(s)).astype(np.float32)
        pts = (qo * np.linalg.norm(s)) @ R.T + t                                                   # observed points
        pts = pts.astype(np.float32)
        ex, ey, ez = rng.uniform(0.8, 1.2, 3)
        aug_bb = np.array([ex, ey, ez], dtype=np.float32)
        aug_rt_t = (rng.uniform(-50, 50, 3) / 1000.0).astype(np.float32)
        aug_rt_r = ref_aug.get_rotation(*rng.uniform(-15, 15, 3))
        seed = 1000 + ci
        prop, bc, nl, noise = replay_draws(args, obj, n, seed)
        tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).clone()
        sym = torch.tensor(SYM[obj], dtype=torch.int32).long()
        torch.manual_seed(seed)
        out = ref_aug.data_augment(args, tt(pts), tt(R), tt(t), tt(s), sym, tt(aug_bb), tt(aug_rt_t), tt(aug_rt_r),
                                   tt(model), 1.0, tt(qo), torch.tensor([obj]).long())
        vals_in = (pts, R, t, s, np.array(SYM[obj], np.int64), aug_bb, aug_rt_t, aug_rt_r, model, qo, obj,
                   np.array([args.aug_bb_pro, args.aug_rt_pro, args.aug_bc_pro, args.aug_pc_pro, args.aug_nl_pro], np.float32),
                   args.aug_pc_r, prop.numpy(), bc.numpy(), nl.numpy(), noise.numpy(), seed)
        for k, v in zip(keys_in, vals_in):
            store[k].append(np.asarray(v))
        for k, v in zip(keys_out, out):
            store[k].append(v.detach().reshape(-1, 3).numpy().copy() if k != "out_t" and k != "out_s"
                            else v.detach().reshape(3).numpy().copy())
    aug = {k: np.stack(v) for k, v in store.items()}
    branches = np.stack([aug["prop"][:, i] < aug["probs"][:, i] for i in range(5)], 1)
    print("data_augment cases:", len(cases), "branch counts (bb, rt, bc*, pc, nl*):", branches.sum(0).tolist(),
          "(* before the class filter)")
    np.savez_compressed(os.path.join(HERE, "data_augment.npz"), **aug)

    # ---- get_bbox ----
    k = 4000
    y1 = rng.integers(-20, 470, k)
    x1 = rng.integers(-20, 630, k)
    hgt = rng.integers(1, 500, k)
    wid = rng.integers(1, 660, k)
    boxes = np.stack([y1, x1, np.minimum(y1 + hgt, 500), np.minimum(x1 + wid, 660)], 1).astype(np.int64)
    boxes[:8] = [[0, 0, 480, 640], [0, 0, 1, 1], [479, 639, 480, 640], [100, 100, 140, 140], [100, 100, 139, 139],
                 [0, 600, 480, 640], [440, 0, 480, 640], [200, 300, 201, 301]]
    wins = np.array([ref_du.get_bbox(b) for b in boxes], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "get_bbox.npz"), boxes=boxes, windows=wins)
    print("get_bbox cases:", k, "window sizes:", sorted(set((wins[:, 1] - wins[:, 0]).tolist())))


if __name__ == "__main__":
    main()
