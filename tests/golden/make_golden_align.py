"""Golden vectors for istnet_amd.align from the REFERENCE's own utils/align.py (numpy only; build container only).

For every synthetic instance: np.random.seed(k), then the reference's estimateSimilarityTransform(source, target).
Stored: the point sets, the seed, and the reference outputs (scale, rotation, translation, transform), plus a flag
for the instances where the reference returns None.  Only data is stored.

    python tests/golden/make_golden_align.py
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def instance(rng, n, outlier_frac, noise):
    q, r = np.linalg.qr(rng.standard_normal((3, 3)))
    q = q * np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    s = rng.uniform(0.1, 0.6)
    t = rng.uniform(-0.5, 0.5, 3) + np.array([0, 0, 0.8])
    src = rng.uniform(-0.5, 0.5, (n, 3)) * rng.uniform(0.3, 1.0, 3)          # NOCS-like coordinates
    tgt = s * src @ q.T + t + rng.standard_normal((n, 3)) * noise
    k = int(outlier_frac * n)
    if k:
        bad = rng.choice(n, k, replace=False)
        tgt[bad] = rng.uniform(-0.5, 0.5, (k, 3)) + t                         # gross outliers (wrong depth / mask)
    return src, tgt


def main():
    sys.path.insert(0, os.path.join(REF, "utils"))
    import align as ref_align
    rng = np.random.default_rng(7)
    n = 256
    cases = [(0.0, 1e-4), (0.1, 1e-3), (0.3, 2e-3), (0.5, 1e-3), (0.7, 1e-3), (0.95, 1e-3), (0.2, 0.0), (0.97, 1e-3)]
    src_all, tgt_all, seeds, outs, oks = [], [], [], [], []
    for k, (frac, noise) in enumerate(cases * 2):
        src, tgt = instance(rng, n, frac, noise)
        seed = 100 + k
        np.random.seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            scale, rot, trans, tf = ref_align.estimateSimilarityTransform(src, tgt)
        src_all.append(src); tgt_all.append(tgt); seeds.append(seed)
        oks.append(scale is not None)
        outs.append(np.full((4, 4), np.nan) if scale is None else tf)
    out = {"source": np.stack(src_all), "target": np.stack(tgt_all), "seeds": np.array(seeds), "ok": np.array(oks),
           "transform": np.stack(outs)}
    # plain Umeyama on a few subsets (no RANSAC)
    um = []
    for i in range(4):
        hom = lambda p: np.vstack([p.T, np.ones((1, p.shape[0]))])
        s, r, t, tf = ref_align.estimateSimilarityUmeyama(hom(out["source"][i][:64]), hom(out["target"][i][:64]))
        um.append(tf)
    out["umeyama64"] = np.stack(um)
    np.savez_compressed(os.path.join(HERE, "align.npz"), **out)
    print({k: v.shape for k, v in out.items()}, "ok:", out["ok"].tolist())


if __name__ == "__main__":
    main()
