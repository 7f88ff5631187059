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


def get_bbox(bbox, img_height=480, img_length=640):
    """utils/data_utils.py:43-71 statement by statement: the square crop window (a multiple of 40 pixels, at most 440)
    centred on the detection box (y1, x1, y2, x2), pushed back inside the 480 x 640 image."""
    y1, x1, y2, x2 = (int(v) for v in bbox)
    window_size = (max(y2 - y1, x2 - x1) // 40 + 1) * 40
    window_size = min(window_size, 440)
    center = [(y1 + y2) // 2, (x1 + x2) // 2]
    rmin = center[0] - int(window_size / 2)
    rmax = center[0] + int(window_size / 2)
    cmin = center[1] - int(window_size / 2)
    cmax = center[1] + int(window_size / 2)
    if rmin < 0:
        delt = -rmin
        rmin = 0
        rmax += delt
    if cmin < 0:
        delt = -cmin
        cmin = 0
        cmax += delt
    if rmax > img_height:
        delt = rmax - img_height
        rmax = img_height
        rmin -= delt
    if cmax > img_length:
        delt = cmax - img_length
        cmax = img_length
        cmin -= delt
    return rmin, rmax, cmin, cmax


def _linear_coeffs(ssize, dsize):
    """Source index and the two 11-bit fixed-point weights of every destination coordinate, as OpenCV's resize() prepares
    them for INTER_LINEAR on 8-bit images (imgproc/resize.cpp: fx = (float)((dx + 0.5) * scale - 0.5), sx = cvFloor(fx),
    taps clamped at both borders, weights saturate_cast<short>(w * INTER_RESIZE_COEF_SCALE), i.e. round-half-to-even)."""
    scale = ssize / dsize                                         # double, as cv::resize computes inv_scale
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= ssize - 1
    f[hi], s[hi] = 0.0, ssize - 1
    w1 = np.rint(f * np.float32(2048.0)).astype(np.int64)         # np.rint: half to even, like cvRound
    w0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
    return s, np.minimum(s + 1, ssize - 1), w0, w1


def resize_linear_u8(img, dsize):
    """cv2.resize(img, (dsize, dsize), interpolation=cv2.INTER_LINEAR) for an (h, w, c) uint8 image, the generic C++ path:
    horizontal pass in int32 (S[x0] * a0 + S[x1] * a1, weights scaled by 2^11), vertical pass
    ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.  PARITY UNPINNED (cv2 is absent from this image): the
    restatement follows OpenCV 4's published source; hardware-specific back ends (IPP, OpenCL) may differ in the last bit."""
    h, w, _ = img.shape
    sx, sx1, a0, a1 = _linear_coeffs(w, dsize)
    sy, sy1, b0, b1 = _linear_coeffs(h, dsize)
    src = img.astype(np.int64)
    rows = src[:, sx, :] * a0[None, :, None] + src[:, sx1, :] * a1[None, :, None]           # (h, dsize, c)
    r0, r1 = rows[sy], rows[sy1]                                                              # (dsize, dsize, c)
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def crop_resize_normalize(image, bbox, img_size=192, reverse_channels=True,
                          mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """provider/dataset.py:213-219 without the colour jitter (the TestDataset path :397-399 has none): channel reversal of
    the BGR image cv2.imread returns, crop [rmin:rmax, cmin:cmax], bilinear resize to img_size, ToTensor (uint8 HWC ->
    float32 CHW / 255) and Normalize ((x - mean) / std in float32).  Returns (resized uint8 (S, S, 3), tensor (3, S, S))."""
    rmin, rmax, cmin, cmax = bbox
    rgb = image[:, :, ::-1] if reverse_channels else image
    crop = rgb[rmin:rmax, cmin:cmax, :]
    small = resize_linear_u8(crop, img_size)
    t = small.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)
    t = (t - np.float32(mean)[:, None, None]) / np.float32(std)[:, None, None]
    return small, t.astype(np.float32)
