"""Similarity-transform estimation between corresponding point sets: Umeyama + RANSAC, batched.

Counterpart of the reference's ``utils/align.py:10-101`` (``estimateSimilarityUmeyama``,
``estimateSimilarityTransform``), which NOCS-style pipelines use to recover (scale, R, t) from predicted NOCS
coordinates and back-projected depth (``data_processing.py:12,182``).  The reference handles ONE instance per call
in numpy and runs its 128 RANSAC hypotheses in a Python loop (5-point Umeyama, residuals, inlier count per
iteration).  Here a whole batch of instances and all hypotheses of every instance are evaluated as a few batched
float64 tensor expressions on whatever device the points live on -- 128 x B three-by-three SVDs in one call,
residual tables (B, 128, N) in one broadcast -- and the reference's sequential semantics (first strictly better
hypothesis wins, early exit once ``1 - (1 - best^5)^i > confidence``) are replayed on the (B, 128) table of inlier
ratios, so the result equals the loop's for the same random draws.

Random draws: the reference calls ``np.random.randint(n, size=5)`` once per iteration.  ``rand_idx`` (B, 128, 5)
carries the draws explicitly (``draw_indices`` produces them from numpy seeds exactly as the reference would consume
them), which is what makes the result reproducible and testable against the reference.
"""
import numpy as np
import torch

MAX_ITER = 128          # ref align.py:57
CONFIDENCE = 0.99       # ref align.py:58
MIN_INLIER_RATIO = 0.1  # ref align.py:87


def _masked_umeyama(src, tgt, mask=None):
    """src, tgt (..., N, 3) float64, mask (..., N) bool or None -> scale (...), rotation (..., 3, 3),
    translation (..., 3), transform (..., 4, 4) mapping src onto tgt.  [ref align.py:10-42]"""
    if mask is None:
        w = torch.ones(src.shape[:-1], dtype=src.dtype, device=src.device)
    else:
        w = mask.to(src.dtype)
    n = w.sum(dim=-1, keepdim=True).clamp_min(1.0)                      # points used per problem
    wn = (w / n).unsqueeze(-1)
    src_mean = (src * wn).sum(dim=-2)
    tgt_mean = (tgt * wn).sum(dim=-2)
    cs = (src - src_mean.unsqueeze(-2)) * w.unsqueeze(-1)
    ct = (tgt - tgt_mean.unsqueeze(-2)) * w.unsqueeze(-1)
    cov = ct.transpose(-1, -2) @ cs / n.unsqueeze(-1)                  # CenteredTarget . CenteredSource^T / n   :17
    u, d, vh = torch.linalg.svd(cov)
    flip = (torch.linalg.det(u) * torch.linalg.det(vh)) < 0.0            # :24
    sign = torch.where(flip, -1.0, 1.0).to(src.dtype)
    d = torch.cat([d[..., :2], d[..., 2:] * sign.unsqueeze(-1)], dim=-1)
    u = torch.cat([u[..., :, :2], u[..., :, 2:] * sign.unsqueeze(-1).unsqueeze(-1)], dim=-1)
    rot = u @ vh                                                         # :29
    var_src = (cs * cs).sum(dim=(-1, -2)) / n.squeeze(-1)                # population variance summed over x, y, z   :31
    scale = d.sum(dim=-1) / var_src                                      # :32
    trans = tgt_mean - scale.unsqueeze(-1) * (rot @ src_mean.unsqueeze(-1)).squeeze(-1)   # :34
    out = torch.zeros(src.shape[:-2] + (4, 4), dtype=src.dtype, device=src.device)
    out[..., :3, :3] = scale.unsqueeze(-1).unsqueeze(-1) * rot
    out[..., :3, 3] = trans
    out[..., 3, 3] = 1.0
    return scale, rot, trans, out


def umeyama(source, target, mask=None):
    """Least-squares similarity transform of corresponding points: source, target (..., N, 3), optional ``mask``
    (..., N) selecting the points of each problem.  Returns (scale, rotation, translation, transform) with
    ``target ~ scale * rotation @ source + translation``; float64.  [ref estimateSimilarityUmeyama, align.py:10-42]"""
    src = torch.as_tensor(source).to(torch.float64)
    tgt = torch.as_tensor(target).to(torch.float64).to(src.device)
    if torch.isnan(src).any() or torch.isnan(tgt).any():
        raise RuntimeError("There are NANs in the input.")                # ref :18-22
    return _masked_umeyama(src, tgt, None if mask is None else torch.as_tensor(mask).to(src.device).bool())


def draw_indices(n_points, seeds, max_iter=MAX_ITER):
    """The 5-point samples the reference's RANSAC loop would draw for one instance per seed:
    ``np.random.seed(seed)`` followed by ``np.random.randint(n_points, size=5)`` once per iteration (align.py:66).
    Returns an int64 array (len(seeds), max_iter, 5)."""
    out = np.empty((len(seeds), max_iter, 5), dtype=np.int64)
    for b, seed in enumerate(seeds):
        rs = np.random.RandomState(seed)
        for i in range(max_iter):
            out[b, i] = rs.randint(n_points, size=5)
    return out


def estimate_similarity_transform(source, target, rand_idx=None, seeds=None, max_iter=MAX_ITER, confidence=CONFIDENCE):
    """RANSAC + Umeyama for a batch of instances.  [ref estimateSimilarityTransform, align.py:45-101]

    source, target: (B, N, 3) corresponding points (or (N, 3) for one instance).
    rand_idx: (B, max_iter, 5) integer samples, or None to draw them with ``draw_indices(N, seeds)`` (seeds default to
    0 .. B-1).
    Returns (scale (B,), rotation (B,3,3), translation (B,3), transform (B,4,4), ok (B,) bool, info dict); rows with
    ``ok == False`` are the instances for which the reference returns ``None`` (best inlier ratio < 0.1) and hold NaN.
    float64 throughout, on the device of ``source``."""
    src = torch.as_tensor(source).to(torch.float64)
    single = src.dim() == 2
    if single:
        src = src.unsqueeze(0)
    tgt = torch.as_tensor(target).to(torch.float64).to(src.device)
    tgt = tgt.unsqueeze(0) if tgt.dim() == 2 else tgt
    if src.shape != tgt.shape or src.shape[-1] != 3:
        raise AssertionError("Source and Target must have same number of points.")       # ref :48
    b, n, _ = src.shape
    dev = src.device
    if rand_idx is None:
        rand_idx = draw_indices(n, list(range(b)) if seeds is None else list(seeds), max_iter)
    ridx = torch.as_tensor(rand_idx).to(dev).long()
    if ridx.dim() == 2:
        ridx = ridx.unsqueeze(0)
    if tuple(ridx.shape) != (b, max_iter, 5):
        raise ValueError(f"rand_idx must have shape ({b}, {max_iter}, 5)")
    # inlier threshold from the source's extent: a tenth of its diameter  [ref :51-56]
    centred = src - src.mean(dim=1, keepdim=True)
    inlier_t = 2.0 * centred.norm(dim=2).amax(dim=1) / 10.0                                   # (B,)
    # all hypotheses at once: 5-point Umeyama per (instance, iteration)  [ref :66-67]
    pick = ridx.reshape(b, max_iter * 5, 1).expand(-1, -1, 3)
    s5 = torch.gather(src, 1, pick).reshape(b, max_iter, 5, 3)
    t5 = torch.gather(tgt, 1, pick).reshape(b, max_iter, 5, 3)
    scale_h, _, _, tf_h = _masked_umeyama(s5, t5)                                             # (B,I), (B,I,4,4)
    # residual of every point under every hypothesis  [ref :68-72]
    mapped = torch.einsum("bhij,bnj->bhni", tf_h[..., :3, :3], src) + tf_h[..., :3, 3].unsqueeze(2)
    resid = (tgt.unsqueeze(1) - mapped).norm(dim=3)                                           # (B,I,N)
    inlier = resid < (scale_h * inlier_t.unsqueeze(1)).unsqueeze(2)                           # NaN hypotheses: no inliers
    ratio = inlier.sum(dim=2).to(torch.float64) / n                                           # (B,I)
    # the loop's bookkeeping on the ratio table: running best, early exit, first strictly better hypothesis [ref :74-84]
    best_so_far = torch.cummax(ratio, dim=1).values
    it = torch.arange(max_iter, device=dev, dtype=torch.float64).unsqueeze(0)
    stop = (1.0 - (1.0 - best_so_far ** 5) ** it) > confidence
    last = torch.where(stop.any(dim=1), stop.to(torch.int64).argmax(dim=1), torch.full((b,), max_iter - 1, device=dev))
    ran = torch.arange(max_iter, device=dev).unsqueeze(0) <= last.unsqueeze(1)                # iterations the loop executes
    masked_ratio = torch.where(ran, ratio, torch.full_like(ratio, -1.0))
    best_ratio, best_it = masked_ratio.max(dim=1)                                             # argmax = FIRST maximum
    first = (masked_ratio == best_ratio.unsqueeze(1)).to(torch.int64).argmax(dim=1)
    ok = best_ratio >= MIN_INLIER_RATIO                                                       # ref :87-89
    best_mask = torch.gather(inlier, 1, first.view(b, 1, 1).expand(-1, 1, n)).squeeze(1)      # (B,N)
    best_mask = torch.where(ok.unsqueeze(1), best_mask, torch.ones_like(best_mask))           # keep failed rows finite
    scale, rot, trans, tf = _masked_umeyama(src, tgt, best_mask)                              # refit on the inliers  [ref :91-93]
    nan = torch.full((), float("nan"), dtype=torch.float64, device=dev)
    scale = torch.where(ok, scale, nan)
    rot = torch.where(ok.view(b, 1, 1), rot, nan)
    trans = torch.where(ok.view(b, 1), trans, nan)
    tf = torch.where(ok.view(b, 1, 1), tf, nan)
    info = {"best_inlier_ratio": best_ratio, "best_iteration": first, "iterations_run": last + 1, "inliers": best_mask}
    if single:
        return scale[0], rot[0], trans[0], tf[0], ok[0], info
    return scale, rot, trans, tf, ok, info
