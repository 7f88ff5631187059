"""istnet_amd.evaluation (3-D IoU, greedy matching, AP, mAP accumulation) against golden vectors produced by the
reference's own utils/evaluation_utils.py (tests/golden/make_golden_map.py)."""
import os

import numpy as np
import pytest
import torch

import istnet_amd  # noqa: F401
from istnet_amd import evaluation

GOLD = os.path.join(os.path.dirname(__file__), "golden", "map_eval.npz")
NAMES = ["BG", "bottle", "bowl", "camera", "can", "laptop", "mug"]
KEYS = ("gt_class_ids", "gt_RTs", "gt_scales", "gt_handle_visibility", "pred_class_ids", "pred_RTs", "pred_scales",
        "pred_scores", "pred_bboxes")


def _results(z):
    return [{k: z[f"im{i}_{k}"] for k in KEYS} for i in range(int(z["n_images"]))]


def _check(device):
    z = np.load(GOLD)
    res = _results(z)
    iou_aps, pose_aps = evaluation.mean_average_precision(
        res, NAMES, degree_thresholds=z["deg"].tolist(), shift_thresholds=z["cm"].tolist(),
        iou_3d_thresholds=z["iou"].tolist(), iou_pose_thres=0.1, device=device)
    assert iou_aps.shape == z["iou_aps"].shape and pose_aps.shape == z["pose_aps"].shape
    np.testing.assert_allclose(iou_aps, z["iou_aps"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(pose_aps, z["pose_aps"], rtol=1e-12, atol=1e-12)
    assert 0.05 < z["iou_aps"][-1].min() and z["pose_aps"][-1, -1, -1] > 0.9        # a non-degenerate fixture
    # the IoU table of one image, including the symmetric classes and the reference's corner-wise extents
    r0 = res[0]
    order = z["im0_order"]
    table = evaluation.iou_3d(torch.as_tensor(r0["pred_RTs"][order], device=device), torch.as_tensor(r0["pred_scales"][order]),
                              torch.as_tensor(r0["gt_RTs"]), torch.as_tensor(r0["gt_scales"]), r0["gt_handle_visibility"],
                              r0["pred_class_ids"][order].astype(np.int64), r0["gt_class_ids"].astype(np.int64), NAMES)
    assert table.dtype == torch.float32
    np.testing.assert_allclose(table.cpu().numpy(), z["im0_iou_table_sorted"], rtol=2e-6, atol=1e-7)


def test_map_matches_reference_cpu():
    _check("cpu")


@pytest.mark.gpu
def test_map_matches_reference_gpu():
    _check("cuda:0")


def test_average_precision_known_values_and_empty_cases():
    ap = evaluation.average_precision(np.array([0, -1, 1, -1]), np.array([0.9, 0.8, 0.7, 0.6]), np.zeros(2))
    assert abs(ap - (0.5 * 1.0 + 0.5 * (2 / 3))) < 1e-12
    assert evaluation.average_precision(np.zeros(0), np.zeros(0), np.zeros(3)) == 0.0
    gm, pm = evaluation.match_by_pose(np.zeros((0, 2, 2)), np.zeros(0), np.array([1, 1]), [5, 360], [2, 100])
    assert gm.shape == (2, 2, 2) and pm.shape == (2, 2, 0) and (gm == -1).all()
    # strictly-greater IoU rule and class gate
    ov = np.array([[0.5, 0.3], [0.5, 0.6]], dtype=np.float32)
    gm, pm = evaluation.match_by_iou(ov, np.array([1, 1]), np.array([1, 2]), [0.5, 0.25])
    assert pm[0].tolist() == [-1, -1] and pm[1].tolist() == [0, -1]


def test_evaluate_reads_result_pickles(tmp_path):
    """evaluate(path): the per-image result pickles of a test run -> the reference's seven headline numbers."""
    import pickle
    z = np.load(GOLD)
    for i, r in enumerate(_results(z)):
        if i == 2:
            r = dict(r); r.pop("gt_handle_visibility")         # older pickles lack the key: all handles visible
        with open(tmp_path / f"results_{i:04d}.pkl", "wb") as fh:
            pickle.dump(r, fh)
    out = evaluation.evaluate(str(tmp_path))
    assert out["iou_3d_aps"].shape == (8, 101) and out["pose_aps"].shape == (8, 62, 22)
    assert set(out["summary"]) == {"3D IoU at 25", "3D IoU at 50", "3D IoU at 75", "5 degree, 2cm", "5 degree, 5cm",
                                   "10 degree, 2cm", "10 degree, 5cm", "10 degree, 10cm"}
    assert 0.0 < out["summary"]["10 degree, 10cm"] <= 100.0 and out["summary"]["3D IoU at 25"] >= out["summary"]["3D IoU at 75"]
