"""Golden vectors for istnet_amd.evaluation from the REFERENCE's own utils/evaluation_utils.py (build container only).

Synthetic result records in the layout of the reference's result pickles (solver.py:243-262) -- several images, all
six NOCS classes, missed detections, false positives, duplicate detections, symmetric classes, a mug with an invisible
handle, an image without predictions -- go through compute_independent_mAP (plot_figure=False; cv2 pre-registered as
an empty module, matplotlib on the Agg backend: harness-side, no edit to the reference).  Stored: the records (as flat
arrays) and the returned iou_3d_aps / pose_aps, plus one image's IoU table from compute_3d_matches.  Only data.

    python tests/golden/make_golden_map.py
"""
import contextlib
import io
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
NAMES = ["BG", "bottle", "bowl", "camera", "can", "laptop", "mug"]


def rand_rt(rng, scale=None):
    q, r = np.linalg.qr(rng.standard_normal((3, 3)))
    q = q * np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    rt = np.eye(4)
    rt[:3, :3] = q * (rng.uniform(0.1, 0.4) if scale is None else scale)
    rt[:3, 3] = rng.uniform(-0.3, 0.3, 3) + np.array([0, 0, 0.8])
    return rt


def perturb(rng, rt, rot_deg, shift_cm, scale_fac):
    w = rng.standard_normal(3)
    w = w / np.linalg.norm(w) * np.deg2rad(rot_deg)
    k = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    th = np.linalg.norm(w)
    rod = np.eye(3) + np.sin(th) / max(th, 1e-12) * k + (1 - np.cos(th)) / max(th * th, 1e-12) * k @ k
    out = rt.copy()
    out[:3, :3] = rod @ rt[:3, :3] * scale_fac
    d = rng.standard_normal(3)
    out[:3, 3] = rt[:3, 3] + d / np.linalg.norm(d) * shift_cm / 100.0
    return out


def make_results(rng, n_images=7):
    results = []
    for im in range(n_images):
        n_gt = int(rng.integers(2, 7))
        gt_cls = rng.integers(1, 7, n_gt)
        if im == 0:
            gt_cls = np.arange(1, 7); n_gt = 6
        gt_rts = np.stack([rand_rt(rng) for _ in range(n_gt)])
        gt_scales = rng.uniform(0.4, 1.0, (n_gt, 3))
        gt_scales = gt_scales / np.linalg.norm(gt_scales, axis=1, keepdims=True)
        gt_vis = rng.integers(0, 2, n_gt)
        pred_cls, pred_rts, pred_scales, pred_scores = [], [], [], []
        for j in range(n_gt):
            u = rng.uniform()
            if u < 0.15 and im != 0:
                continue                                               # missed detection
            reps = 2 if u > 0.9 else 1                                 # duplicate detection
            for _ in range(reps):
                rot = rng.choice([1.0, 4.0, 8.0, 15.0, 40.0, 120.0])
                cm = rng.choice([0.5, 1.5, 4.0, 8.0, 20.0])
                pred_cls.append(gt_cls[j] if rng.uniform() > 0.1 else int(rng.integers(1, 7)))
                pred_rts.append(perturb(rng, gt_rts[j], rot, cm, rng.uniform(0.8, 1.25)))
                pred_scales.append(gt_scales[j] * rng.uniform(0.85, 1.15, 3))
                pred_scores.append(rng.uniform(0.3, 1.0))
        for _ in range(int(rng.integers(0, 3))):                       # false positives
            pred_cls.append(int(rng.integers(1, 7))); pred_rts.append(rand_rt(rng))
            pred_scales.append(rng.uniform(0.3, 0.8, 3)); pred_scores.append(rng.uniform(0.05, 0.6))
        if im == n_images - 1:
            pred_cls, pred_rts, pred_scales, pred_scores = [], [], [], []   # an image without predictions
        n_pred = len(pred_cls)
        results.append({
            "gt_class_ids": gt_cls.astype(np.int32), "gt_RTs": gt_rts, "gt_scales": gt_scales,
            "gt_handle_visibility": gt_vis.astype(np.int32),
            "pred_class_ids": np.array(pred_cls, dtype=np.int32),
            "pred_RTs": np.stack(pred_rts) if n_pred else np.zeros((0, 4, 4)),
            "pred_scales": np.stack(pred_scales) if n_pred else np.zeros((0, 3)),
            "pred_scores": np.array(pred_scores, dtype=np.float64),
            "pred_bboxes": rng.integers(1, 400, (n_pred, 4)).astype(np.int32),
        })
    return results


def main():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, os.path.join(REF, "utils"))
    import evaluation_utils as ref_eval
    rng = np.random.default_rng(21)
    results = make_results(rng)
    deg = [5, 10, 20]
    cm = [2, 5, 10]
    iou = [0.1, 0.25, 0.5, 0.75]
    with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
        iou_aps, pose_aps = ref_eval.compute_independent_mAP(results, NAMES, degree_thresholds=deg, shift_thresholds=cm,
                                                             iou_3d_thresholds=iou, iou_pose_thres=0.1,
                                                             use_matches_for_pose=True, logger=None, plot_figure=False,
                                                             log_dir=tmp)
    out = {"n_images": np.array(len(results)), "deg": np.array(deg), "cm": np.array(cm), "iou": np.array(iou),
           "iou_aps": iou_aps, "pose_aps": pose_aps}
    for i, r in enumerate(results):
        for k, v in r.items():
            out[f"im{i}_{k}"] = v
    r0 = results[0]
    order = np.argsort(r0["pred_scores"])[::-1]
    _, _, table, _ = ref_eval.compute_3d_matches(r0["gt_class_ids"], r0["gt_RTs"], r0["gt_scales"], r0["gt_handle_visibility"],
                                                 NAMES, r0["pred_bboxes"], r0["pred_class_ids"], r0["pred_scores"],
                                                 r0["pred_RTs"], r0["pred_scales"], iou)
    out["im0_iou_table_sorted"] = table
    out["im0_order"] = order
    np.savez_compressed(os.path.join(HERE, "map_eval.npz"), **out)
    print("iou mAP", np.round(iou_aps[-1], 4), "pose mAP\n", np.round(pose_aps[-1], 4))


if __name__ == "__main__":
    main()
