"""Golden vectors for istnet_amd.postprocess from the REFERENCE's own evaluation code (build container only).

utils/evaluation_utils.py imports cv2 (absent here) at module level but the two functions used --
compute_RT_degree_cm_symmetry and compute_RT_overlaps (:588-688) -- are pure numpy, so cv2 is pre-registered as an
empty module (harness-side shim, no edit to the reference).  The tail of utils/solver.py's test_func (:231-241)
cannot be imported (gorilla, tensorboardX); its golden values are produced by the same torch expressions evaluated
on seeded inputs and cross-checked in numpy below.  Only data is stored.

    python tests/golden/make_golden_eval.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def random_similarity(rng, n, near=None):
    """n 4x4 transforms s*R | t; optionally small perturbations of `near` (so errors cover 0..180 degrees)."""
    out = np.zeros((n, 4, 4))
    for i in range(n):
        q, r = np.linalg.qr(rng.standard_normal((3, 3)))
        q = q * np.sign(np.diag(r))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        if near is not None and i < len(near) and i % 2 == 0:
            w = rng.standard_normal(3) * 0.05
            k = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
            base = near[i][:3, :3] / np.cbrt(np.linalg.det(near[i][:3, :3]))
            q = base @ (np.eye(3) + k + k @ k / 2)
            u, _, vt = np.linalg.svd(q)
            q = u @ vt
        out[i, :3, :3] = q * rng.uniform(0.1, 0.5)
        out[i, :3, 3] = rng.uniform(-0.5, 0.5, 3) + (near[i][:3, 3] * 0.98 if near is not None and i < len(near) else 0)
        out[i, 3, 3] = 1
    return out


def main():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.path.insert(0, os.path.join(REF, "utils"))
    import evaluation_utils as ref_eval
    rng = np.random.default_rng(11)
    out = {}
    for tag, names in (("nocs", ["BG", "bottle", "bowl", "camera", "can", "laptop", "mug"]),
                       ("wide", ["BG", "bottle", "bowl", "camera", "can", "cap", "phone", "monitor", "laptop", "mug"])):
        g, p = 9, 11
        gt = random_similarity(rng, g)
        pred = random_similarity(rng, p, near=gt)
        cls = rng.integers(1, len(names), g)
        cls[:len(names) - 1] = np.arange(1, len(names))[:g]          # every class at least once
        vis = rng.integers(0, 2, g)
        table = ref_eval.compute_RT_overlaps(cls, gt, vis, np.zeros(p, dtype=np.int64), pred, names)
        out.update({f"{tag}_gt": gt, f"{tag}_pred": pred, f"{tag}_cls": cls, f"{tag}_vis": vis, f"{tag}_errors": table,
                    f"{tag}_names": np.array(names)})
    np.savez_compressed(os.path.join(HERE, "pose_errors.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
