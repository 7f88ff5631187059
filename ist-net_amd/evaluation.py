"""NOCS-style evaluation of predicted poses: 3-D box IoU, greedy matching, average precision, mAP accumulation.

Counterpart of the reference's ``utils/evaluation_utils.py`` -- ``compute_3d_iou_new`` (:116-173),
``compute_3d_matches`` (:510-586), ``compute_match_from_degree_cm`` (:690-733),
``compute_ap_from_matches_scores`` (:87-113) and the accumulation loop of ``compute_independent_mAP`` (:735-872,
:885-913) -- which ``test.py`` runs over the result pickles after inference (``evaluate``, :1023-1070).

The reference evaluates one (prediction, ground truth) pair at a time in Python: 8 box corners through two 4x4
transforms, 20 re-evaluations for the axis-symmetric classes, a double loop per image and per class.  Here the tables
of an image are tensor expressions -- all pairs, all 20 symmetry rotations at once, float64 on whatever device the
poses live on -- and only the inherently sequential parts (greedy assignment in score order, the precision envelope)
are loops, over arrays of a handful of instances.  Pose errors come from ``postprocess.pose_errors``.

Faithful details worth knowing (they are the reference's behaviour, pinned by tests/golden/map_eval.npz):
* the box extents are taken with ``np.amax(bbox, axis=0)`` on a (3, 8) corner array (:129-132), i.e. per CORNER over
  x, y, z -- an 8-vector "extent", not the axis-aligned box; the IoU is the ratio of products of those 8-vectors;
* IoU tables are stored in float32 before they are compared with the thresholds (:536);
* a prediction matches a ground truth only for IoU strictly greater than the threshold, and stops at the first
  candidate below it (:566-579); pose matches take the unmatched same-class ground truth with the smallest
  degree + cm sum that is within both thresholds (:710-731).
"""
import math

import numpy as np
import torch

from .postprocess import SYNSET_NAMES, pose_errors

_AXIS_SYMMETRIC = ("bottle", "bowl", "can")      # ref :152
_N_ROT = 20                                      # ref :165


def _corners(scales):
    """(N, 3) box sizes -> (N, 3, 8) corner coordinates of the origin-centred box, in the corner order of get_3d_bbox
    (:38-67).  The halving happens in the dtype the sizes arrive in (the reference divides numpy scalars), then float64."""
    half = (scales / 2).to(torch.float64)
    sx = torch.tensor([1, 1, -1, -1, 1, 1, -1, -1], dtype=torch.float64, device=scales.device)
    sy = torch.tensor([1, 1, 1, 1, -1, -1, -1, -1], dtype=torch.float64, device=scales.device)
    sz = torch.tensor([1, -1, 1, -1, 1, -1, 1, -1], dtype=torch.float64, device=scales.device)
    return torch.stack([half[:, 0:1] * sx, half[:, 1:2] * sy, half[:, 2:3] * sz], dim=1)


def _transform(rts, corners):
    """(..., 4, 4) x (..., 3, 8) -> (..., 3, 8): homogeneous transform and division (:70-84)."""
    hom = torch.cat([corners, torch.ones_like(corners[..., :1, :])], dim=-2)
    out = rts @ hom
    return out[..., :3, :] / out[..., 3:4, :]


def iou_3d(pred_RTs, pred_scales, gt_RTs, gt_scales, gt_handle_visibility, pred_class_ids, gt_class_ids,
           synset_names=SYNSET_NAMES):
    """IoU table (P, G), float32 values in a float64-computed expression, of every prediction against every ground
    truth.  [ref compute_3d_iou_new :116-173 as called from compute_3d_matches :539-544]"""
    pred = torch.as_tensor(pred_RTs).to(torch.float64)
    dev = pred.device
    gt = torch.as_tensor(gt_RTs).to(torch.float64).to(dev)
    p, g = pred.shape[0], gt.shape[0]
    if p == 0 or g == 0:
        return torch.zeros(p, g, dtype=torch.float32, device=dev)
    c1 = _corners(torch.as_tensor(pred_scales).to(dev))                  # (P, 3, 8)
    c2 = _corners(torch.as_tensor(gt_scales).to(dev))                    # (G, 3, 8)
    theta = 2 * math.pi * torch.arange(_N_ROT, dtype=torch.float64, device=dev) / float(_N_ROT)
    rot = torch.zeros(_N_ROT, 4, 4, dtype=torch.float64, device=dev)     # y_rotation_matrix (:159-163)
    rot[:, 0, 0] = torch.cos(theta); rot[:, 0, 2] = torch.sin(theta)
    rot[:, 1, 1] = 1.0
    rot[:, 2, 0] = -torch.sin(theta); rot[:, 2, 2] = torch.cos(theta)
    rot[:, 3, 3] = 1.0
    b1 = _transform((pred[:, None] @ rot[None]), c1[:, None])             # (P, R, 3, 8)
    b2 = _transform(gt, c2)                                              # (G, 3, 8)
    # the reference's extents: max / min over x, y, z of each corner (:129-132, axis=0 of a (3, 8) array)
    max1, min1 = b1.amax(dim=-2), b1.amin(dim=-2)                        # (P, R, 8)
    max2, min2 = b2.amax(dim=-2), b2.amin(dim=-2)                        # (G, 8)
    lo = torch.maximum(min1[:, :, None], min2[None, None])               # (P, R, G, 8)
    hi = torch.minimum(max1[:, :, None], max2[None, None])
    ext = hi - lo
    inter = torch.where(ext.amin(dim=-1) < 0, torch.zeros((), dtype=torch.float64, device=dev), ext.prod(dim=-1))
    union = (max1 - min1).prod(dim=-1)[:, :, None] + (max2 - min2).prod(dim=-1)[None, None] - inter
    iou = inter / union                                                  # (P, R, G)
    names = list(synset_names)
    pc = torch.as_tensor(np.asarray(pred_class_ids)).to(dev).long()
    gc = torch.as_tensor(np.asarray(gt_class_ids)).to(dev).long()
    vis = torch.as_tensor(np.asarray(gt_handle_visibility)).to(dev)
    same = pc[:, None] == gc[None]
    sym_cls = torch.tensor([n in _AXIS_SYMMETRIC for n in names], device=dev)[pc]
    mug = torch.tensor([n == "mug" for n in names], device=dev)[pc]
    symmetric = same & (sym_cls[:, None] | (mug[:, None] & (vis[None] == 0)))            # :152
    best = torch.clamp(iou.amax(dim=1), min=0.0)                         # max over the 20 rotations, max_iou starts at 0
    return torch.where(symmetric, best, iou[:, 0]).to(torch.float32)


def _argsort_desc(v):
    """numpy's ``np.argsort(v)[::-1]`` for the short rows that occur here (insertion sort below 16 elements: stable)."""
    return np.argsort(v, kind="stable")[::-1]


def match_by_iou(overlaps, pred_class_ids, gt_class_ids, iou_thresholds):
    """Greedy matching in prediction order (predictions already sorted by score).  overlaps (P, G) float32.
    Returns gt_matches (T, G), pred_matches (T, P) with -1 for unmatched.  [ref compute_3d_matches :547-584]"""
    ov = np.asarray(overlaps)
    p, g = ov.shape
    thr = np.asarray(list(iou_thresholds), dtype=np.float64)
    pred_m = -np.ones((len(thr), p))
    gt_m = -np.ones((len(thr), g))
    order = [_argsort_desc(ov[i]) for i in range(p)]
    for t, th in enumerate(thr):
        for i in range(p):
            for j in order[i]:
                if gt_m[t, j] > -1:
                    continue
                iou = ov[i, j]
                if iou < th:
                    break
                if pred_class_ids[i] != gt_class_ids[j]:
                    continue
                if iou > th:
                    gt_m[t, j] = i
                    pred_m[t, i] = j
                    break
    return gt_m, pred_m


def match_by_pose(errors, pred_class_ids, gt_class_ids, degree_thresholds, shift_thresholds):
    """errors (P, G, 2) = (degrees, cm).  For every threshold pair, predictions in order take the unmatched same-class
    ground truth with the smallest degree + cm sum that is within both thresholds.  Returns gt_matches (D, S, G),
    pred_matches (D, S, P).  [ref compute_match_from_degree_cm :690-733]"""
    err = np.asarray(errors, dtype=np.float64)
    p, g = len(pred_class_ids), len(gt_class_ids)
    nd, ns = len(degree_thresholds), len(shift_thresholds)
    pred_m = -np.ones((nd, ns, p))
    gt_m = -np.ones((nd, ns, g))
    if p == 0 or g == 0:
        return gt_m, pred_m
    order = [np.argsort(err[i].sum(axis=-1), kind="stable") for i in range(p)]
    same = np.asarray(pred_class_ids)[:, None] == np.asarray(gt_class_ids)[None]
    deg_t = np.asarray(list(degree_thresholds), dtype=np.float64)[:, None]
    cm_t = np.asarray(list(shift_thresholds), dtype=np.float64)[None, :]
    for i in range(p):
        free_pred = np.ones((nd, ns), dtype=bool)                     # this prediction still unmatched at (d, s)
        for j in order[i]:
            if not same[i, j]:
                continue
            ok = free_pred & (gt_m[:, :, j] < 0) & ~(err[i, j, 0] > deg_t) & ~(err[i, j, 1] > cm_t)
            gt_m[:, :, j][ok] = i
            pred_m[:, :, i][ok] = j
            free_pred &= ~ok
    return gt_m, pred_m


def average_precision(pred_match, pred_scores, gt_match):
    """VOC-style AP of one class at one threshold.  [ref compute_ap_from_matches_scores :87-113]"""
    pred_match, pred_scores = np.asarray(pred_match), np.asarray(pred_scores)
    assert pred_match.shape[0] == pred_scores.shape[0]
    order = np.argsort(pred_scores)[::-1]
    hit = pred_match[order] > -1
    precisions = np.cumsum(hit) / (np.arange(len(hit)) + 1)
    recalls = np.cumsum(hit).astype(np.float32) / len(gt_match)
    precisions = np.concatenate([[0], precisions, [0]])
    recalls = np.concatenate([[0], recalls, [1]])
    precisions = np.maximum.accumulate(precisions[::-1])[::-1]        # the decreasing envelope (:104-105)
    idx = np.where(recalls[:-1] != recalls[1:])[0] + 1
    return float(np.sum((recalls[idx] - recalls[idx - 1]) * precisions[idx]))


def mean_average_precision(results, synset_names=SYNSET_NAMES, degree_thresholds=(360,), shift_thresholds=(100,),
                           iou_3d_thresholds=(0.1,), iou_pose_thres=0.1, use_matches_for_pose=True, device="cpu"):
    """``results``: per image a dict with gt_class_ids, gt_RTs, gt_scales, gt_handle_visibility, pred_class_ids,
    pred_scores, pred_RTs, pred_scales, pred_bboxes (the reference's result pickles, solver.py:243-262).
    Returns (iou_3d_aps (C + 1, T), pose_aps (C + 1, D + 1, S + 1)) -- the arrays ``compute_independent_mAP`` returns: row c
    = class c, last row = mean over the classes, thresholds as given plus the catch-all 360 degrees / 100 cm.
    [ref compute_independent_mAP :735-872, :885-913]"""
    names = list(synset_names)
    nc = len(names)
    deg_t = list(degree_thresholds) + [360]
    cm_t = list(shift_thresholds) + [100]
    iou_t = list(iou_3d_thresholds)
    if use_matches_for_pose:
        assert iou_pose_thres in iou_t
    iou_pm = [np.zeros((len(iou_t), 0)) for _ in range(nc)]
    iou_ps = [np.zeros((len(iou_t), 0)) for _ in range(nc)]
    iou_gm = [np.zeros((len(iou_t), 0)) for _ in range(nc)]
    pose_pm = [np.zeros((len(deg_t), len(cm_t), 0)) for _ in range(nc)]
    pose_ps = [np.zeros((len(deg_t), len(cm_t), 0)) for _ in range(nc)]
    pose_gm = [np.zeros((len(deg_t), len(cm_t), 0)) for _ in range(nc)]
    for res in results:
        gt_cls = np.asarray(res["gt_class_ids"]).astype(np.int32)
        gt_rts, gt_scales = np.array(res["gt_RTs"]), np.array(res["gt_scales"])
        gt_vis = np.asarray(res["gt_handle_visibility"])
        pred_cls, pred_scores = np.asarray(res["pred_class_ids"]), np.asarray(res["pred_scores"])
        pred_rts, pred_scales = np.array(res["pred_RTs"]), np.asarray(res["pred_scales"])
        pred_boxes = np.array(res["pred_bboxes"])
        if len(gt_cls) == 0 and len(pred_cls) == 0:
            continue
        for c in range(1, nc):
            gsel = gt_cls == c if len(gt_cls) else np.zeros(0, dtype=bool)
            psel = pred_cls == c if len(pred_cls) else np.zeros(0, dtype=bool)
            c_gt_cls = gt_cls[gsel] if len(gt_cls) else np.zeros(0)
            c_gt_rts = gt_rts[gsel] if len(gt_cls) else np.zeros((0, 4, 4))
            c_gt_scales = gt_scales[gsel] if len(gt_cls) else np.zeros((0, 3))
            c_pred_cls = pred_cls[psel] if len(pred_cls) else np.zeros(0)
            c_pred_scores = pred_scores[psel] if len(pred_cls) else np.zeros(0)
            c_pred_rts = pred_rts[psel] if len(pred_cls) else np.zeros((0, 4, 4))
            c_pred_scales = pred_scales[psel] if len(pred_cls) else np.zeros((0, 3))
            c_pred_boxes = pred_boxes[psel] if len(pred_cls) else np.zeros((0, 4))
            if names[c] != "mug":
                c_vis = np.ones_like(c_gt_cls)                                        # :798-802
            else:
                c_vis = gt_vis[gsel] if len(gt_cls) else np.ones(0)
            # --- 3-D IoU matches, predictions in score order (zero boxes are padding) [ref :521-535]
            order = np.zeros(0, dtype=np.int64)
            if len(c_pred_cls):
                keep = ~np.all(c_pred_boxes == 0, axis=1)
                assert keep.all(), "zero-padded prediction boxes are not expected (trim_zeros asserts the same, :31-33)"
                order = np.argsort(c_pred_scores)[::-1]
                c_pred_cls, c_pred_scores = c_pred_cls[order], c_pred_scores[order]
                c_pred_rts, c_pred_scales, c_pred_boxes = c_pred_rts[order], c_pred_scales[order], c_pred_boxes[order]
            overlaps = iou_3d(torch.as_tensor(c_pred_rts, device=device), torch.as_tensor(c_pred_scales, device=device),
                              torch.as_tensor(c_gt_rts, device=device), torch.as_tensor(c_gt_scales, device=device), c_vis,
                              c_pred_cls.astype(np.int64), c_gt_cls.astype(np.int64), names).cpu().numpy()
            gm, pm = match_by_iou(overlaps, c_pred_cls, c_gt_cls, iou_t)
            iou_pm[c] = np.concatenate((iou_pm[c], pm), axis=-1)
            iou_ps[c] = np.concatenate((iou_ps[c], np.tile(c_pred_scores, (len(iou_t), 1))), axis=-1)
            iou_gm[c] = np.concatenate((iou_gm[c], gm), axis=-1)
            # --- pose matches among the instances matched at iou_pose_thres [ref :829-848]
            if use_matches_for_pose:
                t = iou_t.index(iou_pose_thres)
                pk, gk = pm[t] > -1, gm[t] > -1
                c_pred_cls, c_pred_rts, c_pred_scores = c_pred_cls[pk], c_pred_rts[pk], c_pred_scores[pk]
                c_gt_cls, c_gt_rts, c_vis = c_gt_cls[gk], c_gt_rts[gk], c_vis[gk]
            if len(c_pred_cls) and len(c_gt_cls):
                errs = pose_errors(torch.as_tensor(c_pred_rts, device=device), torch.as_tensor(c_gt_rts, device=device),
                                   torch.as_tensor(c_gt_cls.astype(np.int64)), torch.as_tensor(np.asarray(c_vis)),
                                   names).cpu().numpy()
            else:
                errs = np.zeros((len(c_pred_cls), len(c_gt_cls), 2))
            pgm, ppm = match_by_pose(errs, c_pred_cls, c_gt_cls, deg_t, cm_t)
            pose_pm[c] = np.concatenate((pose_pm[c], ppm), axis=-1)
            pose_ps[c] = np.concatenate((pose_ps[c], np.tile(c_pred_scores, (len(deg_t), len(cm_t), 1))), axis=-1)
            pose_gm[c] = np.concatenate((pose_gm[c], pgm), axis=-1)
    iou_aps = np.zeros((nc + 1, len(iou_t)))
    pose_aps = np.zeros((nc + 1, len(deg_t), len(cm_t)))
    for c in range(1, nc):
        for t in range(len(iou_t)):
            iou_aps[c, t] = average_precision(iou_pm[c][t], iou_ps[c][t], iou_gm[c][t])
        for d in range(len(deg_t)):
            for s in range(len(cm_t)):
                pose_aps[c, d, s] = average_precision(pose_pm[c][d, s], pose_ps[c][d, s], pose_gm[c][d, s])
    iou_aps[-1] = iou_aps[1:-1].mean(axis=0)                                          # :884
    pose_aps[-1] = pose_aps[1:-1].mean(axis=0)                                        # :899
    return iou_aps, pose_aps


def evaluate(path, device="cpu"):
    """mAP over the ``results*.pkl`` files of a test run (what test.py writes, solver.py:243-262) with the threshold
    grids of the reference's ``evaluate`` (:1023-1070): degrees 0..60, 0..10 cm in half-centimetre steps, IoU 0..1 in
    hundredths.  Returns {"iou_3d_aps", "pose_aps", "summary"}; ``summary`` holds the seven numbers the reference logs
    (IoU25 / 50 / 75, 5 deg 2 cm, 5 deg 5 cm, 10 deg 2 cm, 10 deg 5 cm, 10 deg 10 cm), in percent."""
    import glob
    import os
    import pickle
    results = []
    for pkl in sorted(glob.glob(os.path.join(path, "results*.pkl"))):
        with open(pkl, "rb") as fh:
            res = pickle.load(fh)
        for r in (res if isinstance(res, list) else [res]):
            if "gt_handle_visibility" not in r:
                r["gt_handle_visibility"] = np.ones_like(r["gt_class_ids"])
            results.append(r)
    deg = list(range(0, 61, 1))
    cm = [i / 2 for i in range(21)]
    iou = [i / 100 for i in range(101)]
    iou_aps, pose_aps = mean_average_precision(results, SYNSET_NAMES, deg, cm, iou, iou_pose_thres=0.1, device=device)
    deg_l, cm_l = deg + [360], cm + [100]
    pick = lambda d, c: 100.0 * pose_aps[-1, deg_l.index(d), cm_l.index(c)]
    summary = {"3D IoU at 25": 100.0 * iou_aps[-1, iou.index(0.25)], "3D IoU at 50": 100.0 * iou_aps[-1, iou.index(0.5)],
               "3D IoU at 75": 100.0 * iou_aps[-1, iou.index(0.75)], "5 degree, 2cm": pick(5, 2), "5 degree, 5cm": pick(5, 5),
               "10 degree, 2cm": pick(10, 2), "10 degree, 5cm": pick(10, 5), "10 degree, 10cm": pick(10, 10)}
    return {"iou_3d_aps": iou_aps, "pose_aps": pose_aps, "summary": summary}
