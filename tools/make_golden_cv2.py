"""Produce tests/golden/fill_missing_cv2.npz with REAL OpenCV -- run where ``cv2`` and a checkout of the reference exist.

    python tools/make_golden_cv2.py [--reference /path/to/IST-Net]        (default /root/reference)

OpenCV is not in the build image, so the depth completion (utils/data_utils.py:514-540 ``fill_missing`` ->
``fill_in_multiscale``: cv2.dilate, morphologyEx(MORPH_CLOSE), medianBlur, bilateralFilter) and the crop resize
(provider/dataset.py:216,398 ``cv2.resize(..., interpolation=cv2.INTER_LINEAR)``) are compared only with numpy restatements
(oracle/depth_fill_oracle.py, oracle/preproc_oracle.py) and hand-derived known answers: PARITY UNPINNED.  This script closes
the gap: it imports the reference's OWN ``fill_missing`` (unmodified), feeds it the synthetic depth scenes of
tests/test_preprocess.py, resizes random crops with cv2, and stores inputs and outputs.  With the file committed,
``tests/test_preprocess.py::test_restatement_matches_real_opencv_golden`` (CPU) and ``::test_kernels_match_real_opencv_golden``
(GPU) stop skipping and pin both the restatement and the HIP kernels to OpenCV's results.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def depth_scene(seed, h=120, w=160):
    """tests/test_preprocess.py::_depth_scene (kept in sync by test_make_golden_cv2_uses_the_tests_scenes)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    d = 400.0 + 1700.0 * yy / h + 480.0 * xx / w + rng.normal(0, 4, (h, w))
    oh, ow = h // 4, w // 4
    d[h // 3:h // 3 + oh, w // 3:w // 3 + ow] = 700.0 + rng.normal(0, 3, (oh, ow))
    d[rng.random((h, w)) < 0.12] = 0
    for _ in range(8):
        r, c, s = int(rng.integers(10, max(h - 20, 11))), int(rng.integers(0, max(w - 20, 1))), int(rng.integers(3, 14))
        d[r:r + s, c:c + s] = 0
    d[:9, :] = 0
    d[rng.random((h, w)) < 0.002] = 3400.0
    return np.clip(d, 0, 65535).astype(np.uint16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    try:
        import cv2
    except ImportError:
        raise SystemExit("cv2 is not importable here: run this script in an environment with opencv-python installed")
    sys.path.insert(0, os.path.join(args.reference, "utils"))
    import data_utils                                        # the reference's module, unmodified
    cv2.setNumThreads(1)
    depth = np.stack([depth_scene(s, 240, 320) for s in (1, 2, 3, 4)])
    filled = np.stack([np.float32(data_utils.fill_missing(d, 1000.0, 1)) for d in depth])     # the call of dataset.py:195
    rng = np.random.default_rng(11)
    image = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    boxes, resized = [], []
    for side in (40, 80, 120, 160, 200, 280, 440):
        r0, c0 = int(rng.integers(0, 480 - side + 1)), int(rng.integers(0, 640 - side + 1))
        boxes.append((r0, r0 + side, c0, c0 + side))
        resized.append(cv2.resize(image[r0:r0 + side, c0:c0 + side], (192, 192), interpolation=cv2.INTER_LINEAR))
    out = os.path.join(ROOT, "tests", "golden", "fill_missing_cv2.npz")
    np.savez_compressed(out, depth=depth, filled=filled, image=image, crop_box=np.array(boxes), resized=np.stack(resized),
                        cv2_version=np.array(cv2.__version__), build_info=np.array(cv2.getBuildInformation()[:2000]))
    print("wrote", out, "with OpenCV", cv2.__version__)


if __name__ == "__main__":
    main()
