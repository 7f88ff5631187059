"""TEST INFRASTRUCTURE ONLY -- numpy / scipy restatement of the reference's depth completion.

Follows utils/data_utils.py:516-540 (fill_missing) and :357-510 (fill_in_multiscale) statement by statement; the OpenCV
calls are restated from their documented semantics (cv2 is not installed in this image, so nothing can be run against it):
  cv2.dilate / cv2.erode / cv2.morphologyEx(MORPH_CLOSE)   max / min over the structuring element, pixels outside the image
                                                           ignored (morphologyDefaultBorderValue)        -> scipy grey_*
  cv2.medianBlur(img, 5) on float32                         5 x 5 median, BORDER_REPLICATE                -> median_filter 'nearest'
  cv2.bilateralFilter(img, 5, 0.5, 2.0)                     disc of radius 2, BORDER_REFLECT_101, weights
                                                           exp(-d^2 / (2 sigma_s^2)) exp(-dI^2 / (2 sigma_c^2))
PARITY UNPINNED: provider/dataset.py and data_utils.py import cv2 and cannot be imported here; this file is checked against
hand-computed known answers (tests/test_preprocess.py).  The product (istnet_amd.preprocess) never imports this module.
"""
import numpy as np
from scipy import ndimage

FULL = lambda k: np.ones((k, k), bool)


def CROSS(k):
    f = np.zeros((k, k), bool)
    f[k // 2, :] = True
    f[:, k // 2] = True
    return f


def dilate(img, footprint):
    return ndimage.grey_dilation(img, footprint=footprint, mode="constant", cval=-np.inf).astype(np.float32)


def erode(img, footprint):
    return ndimage.grey_erosion(img, footprint=footprint, mode="constant", cval=np.inf).astype(np.float32)


def median_blur5(img):
    return ndimage.median_filter(img, size=5, mode="nearest").astype(np.float32)


def bilateral5(img, sigma_color=0.5, sigma_space=2.0):
    h, w = img.shape
    pad = np.pad(img, 2, mode="reflect")                    # BORDER_REFLECT_101
    num = np.zeros((h, w), np.float32)
    den = np.zeros((h, w), np.float32)
    gc, gs = np.float32(-0.5 / sigma_color ** 2), np.float32(-0.5 / sigma_space ** 2)
    for dr in range(-2, 3):
        for dc in range(-2, 3):
            if dr * dr + dc * dc > 4:
                continue
            u = pad[2 + dr:2 + dr + h, 2 + dc:2 + dc + w]
            wgt = np.exp(np.float32(dr * dr + dc * dc) * gs, dtype=np.float32) * np.exp((u - img) ** 2 * gc, dtype=np.float32)
            num += u * wgt
            den += wgt
    return num / den


def fill_in_multiscale(depth_map, max_depth=8.0):
    """data_utils.py:357-510 with the arguments fill_missing uses (extrapolate False, blur_type 'bilateral')."""
    depths_in = np.float32(depth_map)                                                   # :385
    near = (depths_in > 0.01) & (depths_in <= 1.0)                                      # :388-390
    med = (depths_in > 1.0) & (depths_in <= 2.0)
    far = depths_in > 2.0
    s1 = depths_in.copy()                                                               # :393-396
    valid = s1 > 0.01
    s1[valid] = max_depth - s1[valid]
    d_far = dilate(s1 * far, CROSS(3))                                                  # :399-407
    d_med = dilate(s1 * med, CROSS(5))
    d_near = dilate(s1 * near, CROSS(7))
    s2 = s1.copy()                                                                      # :410-418
    s2[d_far > 0.01] = d_far[d_far > 0.01]
    s2[d_med > 0.01] = d_med[d_med > 0.01]
    s2[d_near > 0.01] = d_near[d_near > 0.01]
    s3 = erode(dilate(s2, FULL(5)), FULL(5))                                            # :421-423 MORPH_CLOSE
    s4 = s3.copy()                                                                      # :426-429
    blurred = median_blur5(s3)
    valid = s3 > 0.01
    s4[valid] = blurred[valid]
    top_mask = np.ones(depths_in.shape, bool)                                           # :432-436
    for col in range(s4.shape[1]):
        top_mask[0:np.argmax(s4[:, col] > 0.01), col] = False
    empty = ~(s4 > 0.01) & top_mask                                                     # :439-440
    dil = dilate(s4, FULL(9))                                                           # :443-445
    s5 = s4.copy()
    s5[empty] = dil[empty]
    top_mask = np.ones(s5.shape, bool)                                                  # :448-461 (extrapolate False)
    top_rows = np.argmax(s5 > 0.01, axis=0)
    for col in range(s5.shape[1]):
        top_mask[0:top_rows[col], col] = False
    s7 = s5.copy()                                                                      # :464-468
    for _ in range(6):
        empty = (s7 < 0.01) & top_mask
        dil = dilate(s7, FULL(5))
        s7[empty] = dil[empty]
    blurred = median_blur5(s7)                                                          # :471-473
    valid = (s7 > 0.01) & top_mask
    s7[valid] = blurred[valid]
    blurred = bilateral5(s7, 0.5, 2.0)                                                  # :481-484 (same `valid`)
    s7[valid] = blurred[valid]
    s8 = s7.copy()                                                                      # :487-490
    v = s8 > 0.01
    s8[v] = max_depth - s8[v]
    return s8


def fill_missing(dpt, cam_scale, scale_2_80m):
    """data_utils.py:516-540 (fill_type 'multiscale')."""
    dpt = dpt / cam_scale * scale_2_80m
    final = fill_in_multiscale(dpt.copy(), max_depth=3.0)
    return final / scale_2_80m * cam_scale
