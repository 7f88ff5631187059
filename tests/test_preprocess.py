"""Input preparation: numpy oracle known answers on CPU, HIP kernel vs oracle bit-exact on GPU."""
import numpy as np
import pytest
import torch

import istnet_amd  # noqa: F401
from oracle import preproc_oracle

K = (591.0125, 590.16775, 322.525, 244.11084)


def _scene(seed, h=480, w=640, kind="u16"):
    rng = np.random.default_rng(seed)
    depth = rng.integers(300, 3000, (h, w)).astype(np.uint16)
    depth[rng.random((h, w)) < 0.05] = 0
    if kind == "f32":
        depth = (depth.astype(np.float32) / 1000.0 * 1 + rng.random((h, w)).astype(np.float32) * 1e-3) / 1 * 1000.0
        assert depth.dtype == np.float32
    return rng, depth


def _instances(rng, count, n, h=480, w=640):
    boxes, chooses = [], []
    for side in rng.choice([40, 80, 120, 160, 200, 280, 440], count):
        rmin, cmin = int(rng.integers(0, h - side + 1)), int(rng.integers(0, w - side + 1))
        boxes.append((rmin, rmin + int(side), cmin, cmin + int(side)))
        chooses.append(rng.integers(0, side * side, n))
    return np.array(boxes), np.stack(chooses)


def test_oracle_known_answers():
    depth = np.zeros((480, 640), dtype=np.uint16)
    depth[100, 200] = 1500
    depth[101, 203] = 800
    bbox = (90, 130, 190, 230)                       # 40 x 40 crop
    choose = np.array([10 * 40 + 10, 11 * 40 + 13, 0])
    pts, out = preproc_oracle.backproject_choose(depth, bbox, choose, K)
    assert pts.dtype == np.float32 and out.dtype == np.int64
    np.testing.assert_array_equal(pts[0], np.float32([(200 - K[2]) * 1.5 / K[0], (100 - K[3]) * 1.5 / K[1], 1.5]))
    np.testing.assert_array_equal(pts[1], np.float32([(203 - K[2]) * 0.8 / K[0], (101 - K[3]) * 0.8 / K[1], 0.8]))
    np.testing.assert_array_equal(pts[2], np.zeros(3, np.float32))
    # 40 -> 192: ratio 4.8; (10,10) -> (48,48), (11,13) -> (52,62)
    np.testing.assert_array_equal(out, [48 * 192 + 48, 52 * 192 + 62, 0])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["u16", "f32"])
@pytest.mark.parametrize("shared", [True, False])
def test_backproject_matches_oracle_bit_exact(kind, shared):
    from istnet_amd import preprocess
    count, n = 5, 1024
    rng, depth = _scene(7, kind=kind)
    images = [depth] if shared else [_scene(20 + i, kind=kind)[1] for i in range(count)]
    boxes, choose = _instances(rng, count, n)
    dev = torch.device("cuda:0")
    d = torch.from_numpy(np.stack(images) if not shared else images[0])
    if kind == "u16":
        d = d.view(torch.int16) if d.dtype != torch.uint16 else d
    pts, out = preprocess.backproject_choose(d.to(dev), torch.from_numpy(boxes), torch.from_numpy(choose), K)
    for i in range(count):
        want_pts, want_out = preproc_oracle.backproject_choose(images[0 if shared else i], tuple(boxes[i]), choose[i], K)
        np.testing.assert_array_equal(pts[i].cpu().numpy().view(np.uint32), want_pts.view(np.uint32))
        np.testing.assert_array_equal(out[i].cpu().numpy(), want_out)
    assert int(out.max()) < 192 * 192 and int(out.min()) >= 0


@pytest.mark.gpu
def test_backproject_edge_cases_and_errors():
    from istnet_amd import preprocess
    dev = torch.device("cuda:0")
    depth = torch.zeros(480, 640, dtype=torch.float32, device=dev)
    pts, out = preprocess.backproject_choose(depth, torch.zeros(0, 4, dtype=torch.int32), torch.zeros(0, 16, dtype=torch.int64))
    assert pts.shape == (0, 16, 3) and out.shape == (0, 16)
    pts, out = preprocess.backproject_choose(depth, torch.tensor([[0, 40, 0, 40]]), torch.zeros(1, 0, dtype=torch.int64))
    assert pts.shape == (1, 0, 3)
    with pytest.raises(RuntimeError):
        preprocess.backproject_choose(depth.cpu(), torch.tensor([[0, 40, 0, 40]]), torch.zeros(1, 4, dtype=torch.int64))
    with pytest.raises(TypeError):
        preprocess.backproject_choose(depth.double(), torch.tensor([[0, 40, 0, 40]]), torch.zeros(1, 4, dtype=torch.int64))
    with pytest.raises(ValueError):
        preprocess.backproject_choose(depth, torch.tensor([[0, 40, 0, 40]]), torch.zeros(2, 4, dtype=torch.int64))


# ---------------------------------------------------------------------------------------------
# fill_missing (utils/data_utils.py:357-540): oracle known answers on CPU, HIP kernels vs oracle on the GPU
# ---------------------------------------------------------------------------------------------
def _depth_scene(seed, h=120, w=160):
    """A synthetic raw depth image (uint16 millimetres): a slanted plane 0.4-2.6 m with an object, 12 % dropouts, holes of
    several sizes, an empty band at the top (no return above the scene) and a few far outliers."""
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


def test_depth_fill_oracle_known_answers():
    from oracle import depth_fill_oracle as dfo
    # the OpenCV primitives as restated: dilation ignores the outside of the image, the cross element has no corners
    img = np.zeros((5, 5), np.float32); img[0, 0] = 2.0; img[2, 2] = 1.0
    np.testing.assert_array_equal(dfo.dilate(img, dfo.CROSS(3)),
                                  np.float32([[2, 2, 0, 0, 0], [2, 0, 1, 0, 0], [0, 1, 1, 1, 0], [0, 0, 1, 0, 0], [0] * 5]))
    np.testing.assert_array_equal(dfo.erode(np.ones((4, 4), np.float32), dfo.FULL(5)), np.ones((4, 4), np.float32))
    ramp = np.arange(49, dtype=np.float32).reshape(7, 7)
    assert dfo.median_blur5(ramp)[3, 3] == 24.0 and dfo.median_blur5(ramp)[0, 0] == 2.0       # replicated border: nine 0s, three 1s, three 2s ...
    flat = np.full((6, 6), 1.25, np.float32)
    np.testing.assert_allclose(dfo.bilateral5(flat), flat, rtol=1e-6)
    # end to end on a constant plane with one hole: the hole is filled with the plane's depth, valid pixels are unchanged
    d = np.full((40, 40), 1200, np.uint16); d[20:23, 20:23] = 0
    out = dfo.fill_missing(d, 1000.0, 1)
    assert out.dtype == np.float32 or out.dtype == np.float64
    np.testing.assert_allclose(out, 1200.0, rtol=1e-5)
    # nothing above the first valid row of a column is invented (top mask), empty columns stay empty
    d = np.full((40, 40), 900, np.uint16); d[:12, :] = 0; d[:, 5] = 0
    out = dfo.fill_missing(d, 1000.0, 1)
    assert (out[:8, 20] == 0).all() and out[30, 20] == pytest.approx(900.0, rel=1e-5)


# ---- hand-derived known answers, one per OpenCV rule the restatement relies on -------------------------------------------
# cv2 is not in this image and the reference holds no vector for these functions, so the row stays PARITY UNPINNED until
# tests/golden/fill_missing_cv2.npz exists (tools/make_golden_cv2.py, run where cv2 is installed; test at the end of this
# block).  What can be pinned here is each primitive against numbers worked out by hand from OpenCV's published source.
def test_opencv_morphology_border_rule_known_answers():
    """imgproc/src/morph.dispatch.cpp: with the default border (BORDER_CONSTANT, morphologyDefaultBorderValue() = DBL_MAX) the
    outside of the image NEVER wins -- dilate pads with the type's minimum, erode with its maximum.  Structuring elements of
    utils/data_utils.py:15-77 (FULL_KERNEL_n = ones, CROSS_KERNEL_n = centre row + centre column)."""
    from oracle import depth_fill_oracle as dfo
    # the reference's CROSS_KERNEL_5 / _7 literally (data_utils.py:33-52)
    np.testing.assert_array_equal(dfo.CROSS(5).astype(np.uint8), [[0, 0, 1, 0, 0], [0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [0, 0, 1, 0, 0], [0, 0, 1, 0, 0]])
    assert dfo.CROSS(7).sum() == 13 and dfo.CROSS(7)[3].all() and dfo.CROSS(7)[:, 3].all() and dfo.FULL(9).all()
    # dilate, cross of arm 3, bright pixel in the corner: it spreads 3 pixels along its row and column only
    img = np.zeros((6, 6), np.float32); img[0, 0] = 3.0
    want = np.zeros((6, 6), np.float32); want[0, :4] = 3.0; want[:4, 0] = 3.0
    np.testing.assert_array_equal(dfo.dilate(img, dfo.CROSS(7)), want)
    # dilate, full 5 x 5, pixel at (1, 4) of a 4 x 6 image: the 5 x 5 box clipped to the image
    img = np.zeros((4, 6), np.float32); img[1, 4] = 2.0
    want = np.zeros((4, 6), np.float32); want[0:4, 2:6] = 2.0
    np.testing.assert_array_equal(dfo.dilate(img, dfo.FULL(5)), want)
    # erode ignores the outside: a constant image is a fixed point, up to and including the border ...
    np.testing.assert_array_equal(dfo.erode(np.full((5, 7), 1.5, np.float32), dfo.FULL(5)), np.full((5, 7), 1.5, np.float32))
    # ... so MORPH_CLOSE (dilate then erode, FULL_KERNEL_5: data_utils.py:421-423) of a lone corner pixel KEEPS it: the dilation
    # makes the block [0:3, 0:3], and the erosion window of (0, 0) clipped to the image is exactly that block (zero padding
    # would erase it); every other pixel sees a zero inside the image
    img = np.zeros((8, 8), np.float32); img[0, 0] = 5.0
    want = np.zeros((8, 8), np.float32); want[0, 0] = 5.0
    np.testing.assert_array_equal(dfo.erode(dfo.dilate(img, dfo.FULL(5)), dfo.FULL(5)), want)
    # closing fills a one-pixel hole of a plane and leaves the plane alone
    img = np.full((9, 9), 2.0, np.float32); img[4, 4] = 0.0
    np.testing.assert_array_equal(dfo.erode(dfo.dilate(img, dfo.FULL(5)), dfo.FULL(5)), np.full((9, 9), 2.0, np.float32))


def test_opencv_median_blur_border_known_answers():
    """imgproc/src/median_blur.dispatch.cpp ("The median filter uses BORDER_REPLICATE internally"): 5 x 5 median of
    I[r, c] = 5 r + c.  Corner (0, 0): rows and columns clamp to (0, 0, 0, 1, 2), the 25 values are nine 0s, three 1s,
    three 2s, then 5, 5, 5, 6, 7, 10, 10, 10, 11, 12 -- the 13th smallest is 2.  (0, 4): columns (2, 3, 4, 4, 4): three 2s,
    three 3s, nine 4s first -> 4.  (4, 4): 12, 13, 14, 14, 14, 17, 18, 19, 19, 19, then three 22s -> 22.  Centre: 12."""
    from oracle import depth_fill_oracle as dfo
    img = np.arange(25, dtype=np.float32).reshape(5, 5)
    out = dfo.median_blur5(img)
    assert (out[0, 0], out[0, 4], out[4, 4], out[2, 2]) == (2.0, 4.0, 22.0, 12.0)
    assert out[4, 0] == 20.0     # rows (2, 3, 4, 4, 4) x cols (0, 0, 0, 1, 2): 10 x3, 11, 12, 15 x3, 16, 17, then nine 20s -> 13th is 20


def test_opencv_bilateral_filter_weights_known_answers():
    """imgproc/src/bilateral_filter.dispatch.cpp, bilateralFilter(src, d=5, sigmaColor=0.5, sigmaSpace=2.0) on float32
    (data_utils.py:481-484): radius = d / 2 = 2; taps are the offsets with sqrt(i^2 + j^2) <= radius (13 of the 25: centre, 4 at
    r^2 = 1, 4 at r^2 = 2, 4 at r^2 = 4); space weight exp(-0.5 r^2 / sigmaSpace^2) = exp(-r^2 / 8); colour weight
    exp(-0.5 dI^2 / sigmaColor^2) = exp(-2 dI^2); border BORDER_DEFAULT = BORDER_REFLECT_101 (copyMakeBorder by radius)."""
    from oracle import depth_fill_oracle as dfo
    taps = [(i, j) for i in range(-2, 3) for j in range(-2, 3) if i * i + j * j <= 4]
    assert len(taps) == 13
    ws = {t: np.exp(-(t[0] ** 2 + t[1] ** 2) / 8.0) for t in taps}
    # (a) a vertical step 1 | 2, pixel in the last column of the 1-side, far from the top and bottom: taps with dc >= 1 see dI = 1
    img = np.ones((9, 10), np.float32); img[:, 5:] = 2.0
    num = sum(ws[t] * (np.exp(-2.0) * 2.0 if t[1] >= 1 else 1.0) for t in taps)
    den = sum(ws[t] * (np.exp(-2.0) if t[1] >= 1 else 1.0) for t in taps)
    assert abs(num / den - 1.0554412) < 2e-7             # worked by hand: 7.8493174 / 7.4370009
    np.testing.assert_allclose(dfo.bilateral5(img)[4, 4], num / den, rtol=2e-6)
    np.testing.assert_allclose(dfo.bilateral5(img)[4, 1], 1.0, rtol=1e-7)     # no tap reaches the step
    # (b) REFLECT_101 at the left border: I[r, c] = 0.1 c; column 0 sees columns (2, 1, 0, 1, 2), not (0, 0, 0, 1, 2)
    img = np.tile(np.arange(8, dtype=np.float32) * np.float32(0.1), (9, 1))
    val = lambda c: 0.1 * abs(c)
    num = sum(ws[t] * np.exp(-2.0 * val(t[1]) ** 2) * val(t[1]) for t in taps)
    den = sum(ws[t] * np.exp(-2.0 * val(t[1]) ** 2) for t in taps)
    assert abs(num / den - 0.0710745) < 2e-7             # worked by hand: 0.7023174 / 9.8814237
    np.testing.assert_allclose(dfo.bilateral5(img)[4, 0], num / den, rtol=1e-5)
    replicate = sum(ws[t] * np.exp(-2.0 * (0.1 * max(t[1], 0)) ** 2) * 0.1 * max(t[1], 0) for t in taps) / \
        sum(ws[t] * np.exp(-2.0 * (0.1 * max(t[1], 0)) ** 2) for t in taps)
    assert abs(replicate - num / den) > 0.02             # the two border rules are far apart here: the test discriminates


def test_opencv_inter_linear_fixed_point_known_answers():
    """imgproc/src/resize.cpp, INTER_LINEAR on 8-bit images (provider/dataset.py:216,398): INTER_RESIZE_COEF_BITS = 11.
    Per destination index: fx = (float)((dx + 0.5) * scale - 0.5), sx = cvFloor(fx), fx -= sx; sx < 0 -> (sx, fx) = (0, 0);
    sx >= ssize - 1 -> (ssize - 1, 0); weights saturate_cast<short>((1 - fx) * 2048), saturate_cast<short>(fx * 2048)
    (cvRound: half to even).  Rows: ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2 with S = x0 * a0 + x1 * a1."""
    # up-scaling 40 -> 192 (scale 0.2083..): dx = 0 clamps; dx = 3: fx = 3.5 / 4.8 - 0.5 = 0.229167 -> 1579 / 469; dx = 191 clamps high
    s0, s1, w0, w1 = preproc_oracle._linear_coeffs(40, 192)
    assert (s0[0], w0[0], w1[0]) == (0, 2048, 0) and (s0[3], s1[3], w0[3], w1[3]) == (0, 1, 1579, 469)
    assert (s0[191], s1[191], w0[191], w1[191]) == (39, 39, 2048, 0) and ((w0 + w1) == 2048).all()
    # down-scaling 440 -> 192 (scale 2.291667): dx = 0: fx = 0.645833 -> 725 / 1323; dx = 1: fx = 2.9375 -> sx = 2, 128 / 1920 exactly
    s0, s1, w0, w1 = preproc_oracle._linear_coeffs(440, 192)
    assert (s0[0], w0[0], w1[0]) == (0, 725, 1323) and (s0[1], s1[1], w0[1], w1[1]) == (2, 3, 128, 1920)
    assert (s0[191], w0[191]) == (438, 2048 - w1[191]) and s1.max() == 439
    # pixels: [[7, 200], [90, 33]] -> 3 x 3 (scale 2/3: taps (0 | .5/.5 | 1) per axis).  Centre: rows 7*1024 + 200*1024 = 211968
    # and 125952; (1024 * (211968 >> 4)) >> 16 = 207, (1024 * (125952 >> 4)) >> 16 = 123; (207 + 123 + 2) >> 2 = 83 (mean 82.5)
    img = np.array([[7, 200], [90, 33]], np.uint8)[:, :, None]
    np.testing.assert_array_equal(preproc_oracle.resize_linear_u8(img, 3)[:, :, 0], [[7, 104, 200], [49, 83, 117], [90, 62, 33]])
    # [0, 100] rows -> width 4 (scale 0.5): fx = -.25 (clamped), .25, .75, 1.25 (clamped high): 0, 25, 75, 100
    img = np.array([[0, 100], [0, 100]], np.uint8)[:, :, None]
    np.testing.assert_array_equal(preproc_oracle.resize_linear_u8(img, 4)[0, :, 0], [0, 25, 75, 100])


def test_make_golden_cv2_uses_the_tests_scenes():
    """tools/make_golden_cv2.py generates its inputs with a copy of _depth_scene: the two must stay the same function."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "make_golden_cv2.py")
    spec = importlib.util.spec_from_file_location("make_golden_cv2", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    np.testing.assert_array_equal(mod.depth_scene(3, 60, 80), _depth_scene(3, 60, 80))


def _cv2_golden():
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fill_missing_cv2.npz")
    if not os.path.exists(path):
        pytest.skip("PARITY UNPINNED for fill_missing / cv2.resize: tests/golden/fill_missing_cv2.npz is absent (OpenCV is not in "
                    "this image).  Run `python tools/make_golden_cv2.py` where cv2 and the reference checkout are available "
                    "and commit the file; this test then pins the restatement and the kernels to real OpenCV output.")
    return np.load(path)


def test_restatement_matches_real_opencv_golden():
    """oracle/depth_fill_oracle.py and oracle/preproc_oracle.py against the REFERENCE's own fill_missing
    (utils/data_utils.py:514-540) and cv2.resize(..., INTER_LINEAR) run with real OpenCV (tools/make_golden_cv2.py)."""
    z = _cv2_golden()
    from oracle import depth_fill_oracle as dfo
    for i in range(len(z["depth"])):
        want = z["filled"][i]
        got = np.float32(dfo.fill_missing(z["depth"][i], 1000.0, 1))
        assert ((got == 0) == (want == 0)).all()
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-3)
    for i in range(len(z["crop_box"])):
        r0, r1, c0, c1 = z["crop_box"][i]
        np.testing.assert_array_equal(preproc_oracle.resize_linear_u8(z["image"][r0:r1, c0:c1], 192), z["resized"][i])


@pytest.mark.gpu
def test_kernels_match_real_opencv_golden():
    z = _cv2_golden()
    from istnet_amd import preprocess
    dev = torch.device("cuda:0")
    got = preprocess.fill_missing(torch.from_numpy(z["depth"].view(np.int16)).to(dev), 1000.0, 1).cpu().numpy()
    np.testing.assert_allclose(got, z["filled"], rtol=1e-5, atol=1e-3)
    boxes = torch.from_numpy(z["crop_box"].astype(np.int64))
    _, small = preprocess.crop_resize_normalize(torch.from_numpy(z["image"]).to(dev), boxes, 192, reverse_channels=False,
                                                return_uint8=True)
    np.testing.assert_array_equal(small.cpu().numpy(), z["resized"])


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(120, 160), (480, 640), (37, 53)])
def test_fill_missing_matches_oracle(shape):
    """istnet_depth_fill_multiscale against the numpy / scipy restatement, pass for pass the same selections (dilations,
    medians: exact) and the bilateral filter in float32: the completed depth within 1e-5 relative, zeros in the same
    places, batch entries independent."""
    from istnet_amd import preprocess
    from oracle import depth_fill_oracle as dfo
    dev = torch.device("cuda:0")
    imgs = [_depth_scene(s, *shape) for s in (1, 2, 3)]
    batch = torch.from_numpy(np.stack(imgs).view(np.int16)).to(dev)
    got = preprocess.fill_missing(batch, 1000.0, 1).cpu().numpy()
    for i, img in enumerate(imgs):
        want = np.float32(dfo.fill_missing(img, 1000.0, 1))
        assert ((got[i] == 0) == (want == 0)).all()
        np.testing.assert_allclose(got[i], want, rtol=1e-5, atol=1e-3)           # millimetres
        filled = (img == 0) & (want > 0)
        assert filled.sum() > 0.5 * (img[10:] == 0).sum()                         # the holes below the top band are closed
    single = preprocess.fill_missing(batch[1], 1000.0, 1).cpu().numpy()
    np.testing.assert_array_equal(single, got[1])
    with pytest.raises(RuntimeError):
        preprocess.fill_missing(batch.cpu(), 1000.0, 1)
    with pytest.raises(NotImplementedError):
        preprocess.fill_missing(batch, 1000.0, 1, fill_type="fast")


@pytest.mark.gpu
def test_fill_missing_float64_and_int32_input_round_once():
    """numpy keeps float64 / int32 depth in float64 through ``dpt / cam_scale * scale_2_80m`` and rounds to
    float32 once (utils/data_utils.py:523-526); the kernel's float64 raw kind does the same.  A cam_scale that is not a power
    of two makes double rounding visible; int32 and float64 of the same values must give the uint16 result bit for bit."""
    from istnet_amd import preprocess
    from oracle import depth_fill_oracle as dfo
    dev = torch.device("cuda:0")
    img = _depth_scene(4, 96, 128)
    u16 = torch.from_numpy(img.view(np.int16)).to(dev)
    ref = preprocess.fill_missing(u16, 997.3, 1)
    for conv in (torch.float64, torch.int32, torch.int64):
        got = preprocess.fill_missing(torch.from_numpy(img.astype(np.int64)).to(dev).to(conv), 997.3, 1)
        assert torch.equal(got, ref), conv
    frac = img.astype(np.float64) + np.where(img > 0, 0.37, 0.0)           # a float64 image that is not exactly float32
    got = preprocess.fill_missing(torch.from_numpy(frac).to(dev), 997.3, 1).cpu().numpy()
    want = np.float32(dfo.fill_missing(frac, 997.3, 1))
    assert ((got == 0) == (want == 0)).all()
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-3)


def test_instance_labels_match_numpy_restatement():
    """preprocess.instance_labels (batched tensor form of provider/dataset.py:236-257) against the per-instance numpy
    restatement: symmetric and asymmetric classes, rotation / size / NOCS coordinates / sRT."""
    from istnet_amd import preprocess
    rng = np.random.default_rng(5)
    b, n = 6, 64
    pts = rng.normal(0, 0.1, (b, n, 3)).astype(np.float32) + np.float32([0, 0, 0.8])
    trans = rng.normal(0, 0.05, (b, 3)) + [0, 0, 0.8]
    rots = np.stack([np.linalg.qr(rng.normal(size=(3, 3)))[0] for _ in range(b)])
    scale = rng.uniform(0.1, 0.4, b)
    sizes = rng.uniform(0.3, 1.0, (b, 3))
    sym = np.array([True, False, True, False, False, True])
    got = preprocess.instance_labels(torch.from_numpy(pts), torch.from_numpy(trans), torch.from_numpy(rots),
                                     torch.from_numpy(scale), torch.from_numpy(sizes), torch.from_numpy(sym))
    for i in range(b):
        r, s, qo, srt = preproc_oracle.instance_labels(pts[i], trans[i], rots[i], scale[i], sizes[i], bool(sym[i]))
        np.testing.assert_allclose(got[0][i].numpy(), r, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(got[1][i].numpy(), s, rtol=1e-6)
        np.testing.assert_allclose(got[2][i].numpy(), qo, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(got[3][i].numpy(), srt, rtol=1e-6, atol=1e-7)
    # a symmetric class: the canonical rotation has no in-plane component left (R[0,2] == R[2,0])
    assert abs(float(got[0][0][0, 2] - got[0][0][2, 0])) < 1e-6


def test_get_bbox_matches_the_scalar_restatement():
    """preprocess.get_bbox (batched integer tensor arithmetic) against the statement-by-statement restatement of
    utils/data_utils.py:43-71, including windows pushed back from every image border and the 440-pixel cap."""
    from istnet_amd import preprocess
    rng = np.random.default_rng(3)
    boxes = []
    for _ in range(4000):
        y1, x1 = int(rng.integers(0, 479)), int(rng.integers(0, 639))
        boxes.append((y1, x1, int(rng.integers(y1 + 1, 480)), int(rng.integers(x1 + 1, 640))))
    boxes += [(0, 0, 479, 639), (0, 0, 10, 10), (470, 630, 479, 639), (0, 600, 30, 639), (200, 0, 479, 50)]
    got = preprocess.get_bbox(torch.tensor(boxes)).tolist()
    for b, g in zip(boxes, got):
        assert tuple(g) == preproc_oracle.get_bbox(b), b
    assert preproc_oracle.get_bbox((100, 200, 180, 330)) == (60, 220, 185, 345)       # hand-computed: window 160
    assert preproc_oracle.get_bbox((0, 0, 50, 60)) == (0, 80, 0, 80)                  # pushed back from the corner


def test_resize_oracle_known_answers():
    """The cv2.INTER_LINEAR restatement: a constant image stays constant, an integer 2x down-sampling averages 2 x 2 blocks
    (taps at .5 / .5: (a + b + c + d + 2) >> 2 in this fixed-point form), an identity resize returns the image, and the result
    is within one grey level of a float bilinear interpolation with half-pixel centres."""
    rng = np.random.default_rng(0)
    flat = np.full((40, 40, 3), 77, np.uint8)
    assert (preproc_oracle.resize_linear_u8(flat, 192) == 77).all()
    img = rng.integers(0, 256, (80, 80, 3), dtype=np.uint8)
    np.testing.assert_array_equal(preproc_oracle.resize_linear_u8(img, 80), img)
    half = preproc_oracle.resize_linear_u8(img, 40)
    blocks = img.astype(np.int64).reshape(40, 2, 40, 2, 3)
    want = ((((1024 * ((blocks[:, 0, :, 0] * 1024 + blocks[:, 0, :, 1] * 1024) >> 4)) >> 16)
             + ((1024 * ((blocks[:, 1, :, 0] * 1024 + blocks[:, 1, :, 1] * 1024) >> 4)) >> 16) + 2) >> 2)
    np.testing.assert_array_equal(half, want.astype(np.uint8))
    assert np.abs(half.astype(np.int64) - np.round(blocks.mean(axis=(1, 3)))).max() <= 1
    big = preproc_oracle.resize_linear_u8(img, 192).astype(np.float64)
    ref = torch.nn.functional.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None].double(), size=(192, 192),
                                          mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(big - ref).max() < 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("shared", [True, False])
def test_crop_resize_normalize_matches_oracle_bit_exact(shared):
    """istnet_crop_resize_normalize against the numpy restatement: resized uint8 crops equal, normalised float32 tensors equal
    to the bit, for windows of every size get_bbox produces (40 ... 440) and both channel orders."""
    from istnet_amd import preprocess
    rng = np.random.default_rng(11)
    count = 7
    imgs = rng.integers(0, 256, (1 if shared else count, 480, 640, 3), dtype=np.uint8)
    boxes, _ = _instances(rng, count, 4)
    dev = torch.device("cuda:0")
    image = torch.from_numpy(imgs[0] if shared else imgs).to(dev)
    for rev in (True, False):
        out, small = preprocess.crop_resize_normalize(image, torch.from_numpy(boxes), 192, reverse_channels=rev,
                                                      return_uint8=True)
        for i in range(count):
            want_small, want = preproc_oracle.crop_resize_normalize(imgs[0 if shared else i], tuple(boxes[i]), 192, rev)
            np.testing.assert_array_equal(small[i].cpu().numpy(), want_small)
            np.testing.assert_array_equal(out[i].cpu().numpy().view(np.uint32), want.view(np.uint32))
    only = preprocess.crop_resize_normalize(image, torch.from_numpy(boxes), 64)
    assert only.shape == (count, 3, 64, 64)
    with pytest.raises(RuntimeError):
        preprocess.crop_resize_normalize(image.cpu(), torch.from_numpy(boxes))
    with pytest.raises(TypeError):
        preprocess.crop_resize_normalize(image.float(), torch.from_numpy(boxes))


# ---- pinned to the reference's own outputs (tests/golden/make_golden_augment.py imports its modules) ----
def _golden(name):
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name))


def test_get_bbox_matches_reference_golden():
    """preprocess.get_bbox AND the oracle's restatement against utils/data_utils.py:43-71 itself on 4 000 boxes, boxes
    touching and crossing the image border included (tests/golden/get_bbox.npz)."""
    from istnet_amd import preprocess
    g = _golden("get_bbox.npz")
    got = preprocess.get_bbox(torch.from_numpy(g["boxes"]))
    assert torch.equal(got.to(torch.int64), torch.from_numpy(g["windows"]))
    for b, w in zip(g["boxes"][:500], g["windows"][:500]):
        assert preproc_oracle.get_bbox(tuple(b)) == tuple(w)


def _augment_against_golden(device):
    from istnet_amd import preprocess
    g = _golden("data_augment.npz")
    T = lambda k: torch.from_numpy(g[k]).to(device)
    probs = g["probs"]
    taken = np.zeros(5, dtype=np.int64)
    for uniq in np.unique(probs, axis=0):             # one batched call per probability setting of the fixture
        idx = torch.from_numpy(np.where((probs == uniq).all(1))[0]).to(device)
        draws = {k: T(k)[idx] for k in ("prop", "bc", "nl", "noise")}
        inputs = [T(k)[idx] for k in ("pts", "R", "t", "s", "sym", "aug_bb", "aug_rt_t", "aug_rt_r", "model", "qo", "obj_id")]
        before = [t.clone() for t in inputs]
        out = preprocess.data_augment(tuple(float(v) for v in uniq), *inputs, pc_r=float(g["pc_r"][0]), draws=draws)
        for a, b in zip(inputs, before):
            assert torch.equal(a, b)                  # (the reference's functions write into their arguments; this one does not)
        for name, o in zip(("out_pts", "out_R", "out_t", "out_s", "out_model", "out_qo"), out):
            assert o.device.type == torch.device(device).type
            torch.testing.assert_close(o.cpu(), torch.from_numpy(g[name])[idx.cpu()], rtol=1e-6, atol=1e-6,
                                       msg=lambda m, name=name, uniq=uniq: f"{name} at probs {uniq}: {m}")
        taken += (g["prop"][idx.cpu().numpy()] < uniq).sum(0)
    assert (taken >= 20).all(), taken                 # every branch of data_augment is exercised by the fixture


def test_data_augment_matches_reference_golden():
    """preprocess.data_augment (batched) against provider/data_augmentation.py:217-285 run sample by sample with the
    same random draws: every branch alone, all together, the shipped probabilities and 0.5 each; all six classes
    (symmetric / asymmetric box deformation, the mug / bowl cage, both non-linear axes).  1e-6."""
    _augment_against_golden("cpu")


@pytest.mark.gpu
def test_data_augment_matches_reference_golden_on_device():
    _augment_against_golden("cuda:0")


def test_data_augment_draws_its_own_numbers():
    """Without ``draws`` the batch is augmented from a torch generator: reproducible, a no-op at probability zero, and
    rigid where only the rigid branch is on (pairwise distances preserved)."""
    from istnet_amd import preprocess
    g = _golden("data_augment.npz")
    T = lambda k: torch.from_numpy(g[k][:12])
    args = [T(k) for k in ("pts", "R", "t", "s", "sym")]
    bb, tr, rm = preprocess.generate_aug_parameters(12, generator=torch.Generator().manual_seed(5))
    assert bb.shape == (12, 3) and float(bb.min()) >= 0.8 and float(bb.max()) < 1.2 and float(tr.abs().max()) <= 0.05
    torch.testing.assert_close(rm @ rm.transpose(1, 2), torch.eye(3).expand(12, 3, 3), rtol=0, atol=1e-6)
    rest = [T("model"), T("qo"), T("obj_id")]
    same = preprocess.data_augment((0.0,) * 5, *args, bb, tr, rm, *rest, generator=torch.Generator().manual_seed(1))
    for got, key in zip(same, ("pts", "R", "t", "s", "model", "qo")):
        assert torch.equal(got, T(key))
    a = preprocess.data_augment((0.0, 1.0, 0.0, 0.0, 0.0), *args, bb, tr, rm, *rest, generator=torch.Generator().manual_seed(1))
    b = preprocess.data_augment((0.0, 1.0, 0.0, 0.0, 0.0), *args, bb, tr, rm, *rest, generator=torch.Generator().manual_seed(1))
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    pair = lambda p: (p[:, :, None] - p[:, None]).norm(dim=-1)
    torch.testing.assert_close(pair(a[0]), pair(T("pts")), rtol=1e-4, atol=1e-6)


def test_jitter_points_is_the_dataset_line():
    """provider/dataset.py:211 restated in numpy (float64 noise on float32 points, one rounding) against the batched form."""
    from istnet_amd import preprocess
    rng = np.random.default_rng(2)
    pts = rng.uniform(-0.3, 0.9, (3, 64, 3)).astype(np.float32)
    z = rng.standard_normal((3, 64, 3)) * np.array([1.0, 4.0, 8.0]).reshape(3, 1, 1)      # some draws beyond the clip
    want = (pts + np.clip(0.001 * z, -0.005, 0.005)).astype(np.float32)
    got = preprocess.jitter_points(torch.from_numpy(pts), noise=torch.from_numpy(z))
    assert got.dtype == torch.float32 and np.array_equal(got.numpy(), want)
    assert float((got - torch.from_numpy(pts)).abs().max()) <= 0.005 + 1e-7
