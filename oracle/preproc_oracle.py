"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's per-instance input preparation.

Follows provider/dataset.py:203-210 (full-image back-projection in numpy's own dtype promotion), :209 (crop +
`choose` gather) and :226-231 (remap of `choose` to the resized crop); the TestDataset block :348-355,392,401-405
is the same arithmetic.  PARITY UNPINNED: provider/dataset.py cannot be imported here (cv2, torchvision) and the
arithmetic sits inline in __getitem__, so this file is checked against hand-computed known answers only
(tests/test_preprocess.py).  The product (istnet_amd.preprocess) never imports this module.
"""
import numpy as np


def backproject_choose(depth, bbox, choose, intrinsics, norm_scale=1000.0, img_size=192):
    """depth (h,w) uint16 or float32, bbox = (rmin,rmax,cmin,cmax), choose (n,) -> pts (n,3) f32, choose_out (n,) i64."""
    h, w = depth.shape
    xmap = np.array([[i for i in range(w)] for j in range(h)])            # dataset.py:58-59 / :312-313
    ymap = np.array([[j for i in range(w)] for j in range(h)])
    cam_fx, cam_fy, cam_cx, cam_cy = intrinsics
    pts2 = depth.copy() / norm_scale                                        # :205
    pts0 = (xmap - cam_cx) * pts2 / cam_fx                                  # :206
    pts1 = (ymap - cam_cy) * pts2 / cam_fy                                  # :207
    pts = np.transpose(np.stack([pts0, pts1, pts2]), (1, 2, 0)).astype(np.float32)   # :208
    rmin, rmax, cmin, cmax = bbox
    pts = pts[rmin:rmax, cmin:cmax, :].reshape((-1, 3))[choose, :]          # :209
    crop_w = rmax - rmin                                                    # :226
    ratio = img_size / crop_w
    col_idx = choose % crop_w
    row_idx = choose // crop_w
    choose_out = (np.floor(row_idx * ratio) * img_size + np.floor(col_idx * ratio)).astype(np.int64)   # :231
    return pts, choose_out
