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


def instance_labels(pts, translation, rotation, scale, sizes, symmetric):
    """provider/dataset.py:236-257 for one instance: rotation canonicalised about the y axis for a symmetric class, size
    label, NOCS coordinates ``qo`` of the sampled points and the 4x4 sRT.  pts (n,3) f32, translation (3,), rotation (3,3),
    scale scalar, sizes (3,)."""
    import math
    translation = translation.astype(np.float32)
    rotation = rotation.astype(np.float32)
    size = scale * sizes.astype(np.float32)                                      # :239
    if symmetric:                                                                # :241-248
        theta_x = rotation[0, 0] + rotation[2, 2]
        theta_y = rotation[0, 2] - rotation[2, 0]
        r_norm = math.sqrt(theta_x ** 2 + theta_y ** 2)
        s_map = np.array([[theta_x / r_norm, 0.0, -theta_y / r_norm],
                          [0.0, 1.0, 0.0],
                          [theta_y / r_norm, 0.0, theta_x / r_norm]])
        rotation = rotation @ s_map
    qo = (pts - translation[np.newaxis, :]) / (np.linalg.norm(size) + 1e-8) @ rotation      # :249
    sRT = np.identity(4, dtype=np.float32)                                       # :251-253
    sRT[:3, :3] = scale * rotation
    sRT[:3, 3] = translation
    return rotation, size, qo, sRT
