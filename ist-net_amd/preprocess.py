"""Per-instance input preparation on the GPU.

``backproject_choose`` replaces the numpy block of the reference's Dataset classes that turns a depth image, an
instance crop and the sampled pixel list into the network inputs ``pts`` and ``choose``
(provider/dataset.py:203-210,226-231 for training, :348-355,392,401-405 for testing).  The host code there builds
the whole (480,640,3) float64 point map for every image; the kernel (csrc/preproc.hip through
include/istnet_preproc.h) back-projects only the sampled pixels, bit-exactly.  ``get_bbox`` (the square crop window),
``crop_resize_normalize`` (RGB crop -> cv2-style bilinear resize -> ToTensor / Normalize), ``fill_missing`` (depth
completion) and ``instance_labels`` cover the rest of the tensor arithmetic of ``__getitem__``; image decoding, the mask
test, the random pixel sampling (`np.random.choice`) and the random colour jitter stay with the caller.
"""
import torch

from . import _native

REAL_INTRINSICS = (591.0125, 590.16775, 322.525, 244.11084)     # fx, fy, cx, cy   [ref dataset.py:305]
CAMERA_INTRINSICS = (577.5, 577.5, 319.5, 239.5)


def backproject_choose(depth, bboxes, choose, intrinsics=REAL_INTRINSICS, norm_scale=1000.0, img_size=192):
    """depth: (h,w) shared by all instances or (count,h,w), uint16 (raw millimetres; int16 storage accepted) or
    float32 (after fill_missing); bboxes (count,4) rmin,rmax,cmin,cmax; choose (count,n) flat crop indices.
    Returns pts (count,n,3) float32 and choose_out (count,n) int64.  CUDA tensors only (no CPU path)."""
    if not depth.is_cuda:
        raise RuntimeError("backproject_choose: CPU not supported")
    if depth.dtype not in (torch.uint16, torch.int16, torch.float32):
        raise TypeError(f"backproject_choose: depth must be uint16 or float32, got {depth.dtype}")
    if depth.dim() not in (2, 3) or choose.dim() != 2 or bboxes.shape != (choose.size(0), 4):
        raise ValueError("backproject_choose: expected depth (h,w) or (count,h,w), bboxes (count,4), choose (count,n)")
    count, n = choose.shape
    if depth.dim() == 3 and depth.size(0) != count:
        raise ValueError("backproject_choose: one depth image per instance, or one shared image")
    depth = depth.contiguous()
    h, w = depth.shape[-2:]
    bboxes = bboxes.to(device=depth.device, dtype=torch.int32).contiguous()
    choose = choose.to(device=depth.device, dtype=torch.int32).contiguous()
    pts = torch.empty(count, n, 3, dtype=torch.float32, device=depth.device)
    out = torch.empty(count, n, dtype=torch.int64, device=depth.device)
    fx, fy, cx, cy = (float(v) for v in intrinsics)
    with torch.cuda.device(depth.device):
        rc = _native.lib().istnet_backproject_choose(
            count, n, h, w, depth.data_ptr(), 1 if depth.dtype == torch.float32 else 0,
            h * w if depth.dim() == 3 else 0, bboxes.data_ptr(), choose.data_ptr(), fx, fy, cx, cy,
            float(norm_scale), int(img_size), pts.data_ptr(), out.data_ptr(),
            torch.cuda.current_stream(depth.device).cuda_stream)
    if rc != 0:
        raise RuntimeError(f"istnet_backproject_choose failed with code {rc}")
    return pts, out


IMAGENET_MEAN = (0.485, 0.456, 0.406)       # [ref dataset.py:103-105]
IMAGENET_STD = (0.229, 0.224, 0.225)


def get_bbox(bboxes, img_height=480, img_length=640):
    """utils/data_utils.py:43-71 for a batch: detection boxes (count, 4) = y1, x1, y2, x2 -> square crop windows (count, 4) =
    rmin, rmax, cmin, cmax (side a multiple of 40 pixels, at most 440, centred on the box, pushed back inside the image).
    Integer tensor arithmetic, any device."""
    bb = torch.as_tensor(bboxes).to(torch.int64).reshape(-1, 4)
    y1, x1, y2, x2 = bb.unbind(1)
    window = torch.clamp((torch.maximum(y2 - y1, x2 - x1).div(40, rounding_mode="floor") + 1) * 40, max=440)
    half = window.div(2, rounding_mode="trunc")                   # int(window_size / 2)
    cy, cx = (y1 + y2).div(2, rounding_mode="floor"), (x1 + x2).div(2, rounding_mode="floor")
    rmin, rmax, cmin, cmax = cy - half, cy + half, cx - half, cx + half
    d = torch.clamp(-rmin, min=0); rmin, rmax = rmin + d, rmax + d          # the four `if` blocks, in the reference's order
    d = torch.clamp(-cmin, min=0); cmin, cmax = cmin + d, cmax + d
    d = torch.clamp(rmax - img_height, min=0); rmin, rmax = rmin - d, rmax - d
    d = torch.clamp(cmax - img_length, min=0); cmin, cmax = cmin - d, cmax - d
    return torch.stack([rmin, rmax, cmin, cmax], dim=1).to(torch.int32)


def crop_resize_normalize(image, bboxes, img_size=192, reverse_channels=True, mean=IMAGENET_MEAN, std=IMAGENET_STD,
                          return_uint8=False):
    """provider/dataset.py:213-219 / :397-399 on the GPU: ``image`` (h, w, 3) uint8 shared by all instances or
    (count, h, w, 3), as cv2.imread returns it (BGR: ``reverse_channels``); bboxes (count, 4) rmin, rmax, cmin, cmax.
    Returns the network input (count, 3, img_size, img_size) float32 -- crop, cv2.INTER_LINEAR resize (OpenCV's 8-bit
    fixed-point arithmetic), / 255, Normalize -- and with ``return_uint8`` also the resized crops (count, S, S, 3) uint8
    (what a colour jitter would be applied to).  CUDA tensors only."""
    if not image.is_cuda:
        raise RuntimeError("crop_resize_normalize: CPU not supported")
    if image.dtype != torch.uint8 or image.dim() not in (3, 4) or image.shape[-1] != 3:
        raise TypeError("crop_resize_normalize: image must be uint8 (h, w, 3) or (count, h, w, 3)")
    bboxes = torch.as_tensor(bboxes).to(device=image.device, dtype=torch.int32).reshape(-1, 4).contiguous()
    count = bboxes.size(0)
    if image.dim() == 4 and image.size(0) != count:
        raise ValueError("crop_resize_normalize: one image per instance, or one shared image")
    image = image.contiguous()
    h, w = image.shape[-3], image.shape[-2]
    out = torch.empty(count, 3, img_size, img_size, dtype=torch.float32, device=image.device)
    small = torch.empty(count, img_size, img_size, 3, dtype=torch.uint8, device=image.device) if return_uint8 else None
    import ctypes
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    sd = (ctypes.c_float * 3)(*[float(v) for v in std])
    with torch.cuda.device(image.device):
        rc = _native.lib().istnet_crop_resize_normalize(
            count, h, w, image.data_ptr(), h * w * 3 if image.dim() == 4 else 0, 1 if reverse_channels else 0,
            bboxes.data_ptr(), int(img_size), ctypes.cast(m, ctypes.c_void_p), ctypes.cast(sd, ctypes.c_void_p),
            small.data_ptr() if small is not None else None, out.data_ptr(),
            torch.cuda.current_stream(image.device).cuda_stream)
    if rc != 0:
        raise RuntimeError(f"istnet_crop_resize_normalize failed with code {rc}")
    return (out, small) if return_uint8 else out


def fill_missing(dpt, cam_scale, scale_2_80m, fill_type="multiscale", extrapolate=False, show_process=False,
                 blur_type="bilateral"):
    """Depth completion of the reference's data pipeline (utils/data_utils.py:516-540, same signature; the Dataset
    classes call ``fill_missing(depth, norm_scale, 1)``, provider/dataset.py:172-173,361-362) on the GPU:
    ``dpt`` (h, w) or (b, h, w) raw depth -- a CUDA tensor (uint16 / int16 / float) -- is scaled by
    ``scale_2_80m / cam_scale``, completed by the multi-scale morphological pipeline of fill_in_multiscale
    (istnet_depth_fill_multiscale, csrc/depth_fill.hip) and scaled back; returns float32 of the same shape.
    Only the reference's own configuration is implemented: fill_type 'multiscale', extrapolate False, blur_type
    'bilateral'.  CUDA tensors only (no CPU path)."""
    if fill_type != "multiscale" or extrapolate or blur_type != "bilateral" or show_process:
        raise NotImplementedError("fill_missing: only fill_type='multiscale', extrapolate=False, blur_type='bilateral' "
                                  "(what the reference's Dataset classes use)")
    if not torch.is_tensor(dpt) or not dpt.is_cuda:
        raise RuntimeError("fill_missing: CPU not supported")
    squeeze = dpt.dim() == 2
    img = dpt.unsqueeze(0) if squeeze else dpt
    if img.dim() != 3:
        raise ValueError("fill_missing: expected a depth image (h, w) or a batch (b, h, w)")
    if img.dtype in (torch.uint16, torch.int16):
        raw, is_float = img.contiguous(), 0                             # raw millimetres, 16-bit storage read as unsigned
    elif img.dtype == torch.float32:
        raw, is_float = img.contiguous(), 1
    else:
        # float64 / int32 / int64 ...: numpy scales these in float64 and rounds to float32 ONCE (utils/data_utils.py:523-526);
        # a float32 conversion here would round twice and can move a value across the 0.1 / max_depth thresholds
        raw, is_float = img.to(torch.float64).contiguous(), 2
    b, h, w = raw.shape
    lib = _native.lib()
    scratch = torch.empty(lib.istnet_depth_fill_scratch_floats(b, h, w), dtype=torch.float32, device=raw.device)
    out = torch.empty((b, h, w), dtype=torch.float32, device=raw.device)
    with torch.cuda.device(raw.device):
        # numpy evaluates dpt / cam_scale * scale_2_80m in float64 and fill_in_multiscale rounds to float32 once; the result is
        # scaled back in float32: both inside the kernels (include/istnet_preproc.h)
        _native.check(lib.istnet_depth_fill_missing(b, h, w, raw.data_ptr(), is_float, float(cam_scale), float(scale_2_80m), 3.0,
                                                    scratch.data_ptr(), out.data_ptr(),
                                                    torch.cuda.current_stream(raw.device).cuda_stream), "depth_fill_missing")
    return out[0] if squeeze else out


def instance_labels(pts, translation, rotation, scale, sizes, symmetric):
    """Pose labels of a batch of instances, the arithmetic of provider/dataset.py:236-257 as batched tensor expressions
    (device-agnostic): for classes with a rotational symmetry about the y axis (``symmetric`` (B,) bool: bottle, bowl, can in
    NOCS) the rotation is canonicalised by the in-plane map s_map; ``size = scale * sizes``; the NOCS coordinates of the sampled
    points ``qo = (pts - t) / (|size| + 1e-8) @ R``; ``sRT`` = [scale * R | t].  pts (B,n,3), translation (B,3), rotation
    (B,3,3), scale (B,), sizes (B,3).  Returns rotation (B,3,3), size (B,3), qo (B,n,3) -- float64 like the numpy code, which
    promotes through s_map / np.linalg.norm; the caller casts as the reference's ``torch.FloatTensor`` does -- and sRT (B,4,4)
    float32."""
    f64 = torch.float64
    t32 = translation.to(torch.float32)
    rot = rotation.to(torch.float32)
    size = scale.reshape(-1, 1).to(torch.float32) * sizes.to(torch.float32)
    theta_x = (rot[:, 0, 0] + rot[:, 2, 2]).to(f64)
    theta_y = (rot[:, 0, 2] - rot[:, 2, 0]).to(f64)
    r_norm = torch.sqrt(theta_x ** 2 + theta_y ** 2)
    cx, sy = theta_x / r_norm, theta_y / r_norm
    zero, one = torch.zeros_like(cx), torch.ones_like(cx)
    s_map = torch.stack([torch.stack([cx, zero, -sy], 1), torch.stack([zero, one, zero], 1), torch.stack([sy, zero, cx], 1)], 1)
    sym = symmetric.reshape(-1, 1, 1).to(torch.bool)
    rot64 = torch.where(sym, rot.to(f64) @ s_map, rot.to(f64))
    rot_out = torch.where(sym, rot64, rot.to(f64))
    norm = torch.linalg.vector_norm(size, dim=1).to(f64) + 1e-8          # np.linalg.norm of a float32 vector is float32
    qo = ((pts.to(torch.float32) - t32.unsqueeze(1)).to(f64) / norm.reshape(-1, 1, 1)) @ rot_out
    srt = torch.eye(4, dtype=torch.float32, device=pts.device).repeat(pts.shape[0], 1, 1)
    srt[:, :3, :3] = (scale.reshape(-1, 1, 1).to(f64) * rot_out).to(torch.float32)
    srt[:, :3, 3] = t32
    return rot_out, size, qo, srt


_CONSTS = {}


def _const(values, dtype, device):
    """A small constant tensor on ``device``, uploaded once per (values, dtype, device): a host-to-device copy of pageable
    memory per call is what keeps a sequence of tensor expressions from being captured in a HIP graph."""
    key = (tuple(values), dtype, str(device))
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = torch.tensor(list(values), dtype=dtype, device=device)
    return t


def _draw(fn, shape, dtype, device, generator):
    """Random numbers for ``device``: drawn THERE when no generator is given (the device's default generator: no host
    work, no copy, capturable in a HIP graph) or when the generator lives there; a CPU generator (replayable draws, the
    tests) draws on the host and copies."""
    where = device if generator is None else generator.device
    return fn(tuple(shape), generator=generator, dtype=dtype, device=where).to(device)


AUG_PROBS_DEFAULT = (0.3, 0.3, 0.0, 0.0, 0.0)     # aug_bb_pro, aug_rt_pro, aug_bc_pro, aug_pc_pro, aug_nl_pro [ref config/ist_net_default.yaml:37-42]


def generate_aug_parameters(count, device="cpu", generator=None, s_x=(0.8, 1.2), s_y=(0.8, 1.2), s_z=(0.8, 1.2), ax=50,
                            ay=50, az=50, a=15):
    """provider/dataset.py:123-133 for a batch: box stretch factors (count, 3) in [0.8, 1.2), a translation (count, 3) of up
    to +-50 mm per axis (in metres) and a rotation (count, 3, 3) of up to +-15 degrees about each axis, R_z R_y R_x as
    data_augmentation.get_rotation (:8-25) composes it.  Drawn from ``generator`` (torch), not from numpy's global state."""
    import math
    u = _draw(torch.rand, (count, 9), torch.float64, torch.device(device), generator)
    lo = _const([s_x[0], s_y[0], s_z[0]], torch.float64, device)
    hi = _const([s_x[1], s_y[1], s_z[1]], torch.float64, device)
    bb = (u[:, 0:3] * (hi - lo) + lo).to(torch.float32)
    ang = (u[:, 3:6] * 2 * a - a) / 180.0 * math.pi
    lim = _const([ax, ay, az], torch.float64, device)
    trans = ((u[:, 6:9] * 2 * lim - lim).to(torch.float32) / 1000.0)
    cx, cy, cz = torch.cos(ang).unbind(1)
    sx, sy, sz = torch.sin(ang).unbind(1)
    one, zero = torch.ones_like(cx), torch.zeros_like(cx)
    rx = torch.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx], 1).view(-1, 3, 3)
    ry = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], 1).view(-1, 3, 3)
    rz = torch.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one], 1).view(-1, 3, 3)
    return bb, trans, (rz @ ry @ rx).to(torch.float32)


def data_augment(probs, pts, rotation, translation, size, sym, aug_bb, aug_rt_t, aug_rt_r, model, nocs, obj_id, pc_r=0.002,
                 draws=None, generator=None):
    """The point-cloud augmentation of the reference's training set for a whole batch at once
    (provider/data_augmentation.py:217-285 ``data_augment``, called per sample from provider/dataset.py:277-283), as
    batched tensor expressions on whatever device the inputs live on -- the reference walks the samples one by one on
    the host inside ``__getitem__``.

    ``probs``: (aug_bb_pro, aug_rt_pro, aug_bc_pro, aug_pc_pro, aug_nl_pro) -- object with those attributes (the
    reference's config) or a 5-sequence.  pts (B, N, 3), rotation (B, 3, 3), translation (B, 3), size (B, 3), sym
    (B, 4) integer symmetry codes (only column 0 is read, :49), aug_bb / aug_rt_t (B, 3), aug_rt_r (B, 3, 3)
    (``generate_aug_parameters``), model (B, M, 3), nocs (B, N, 3), obj_id (B,) 0-based class ids.
    Returns (pts, rotation, translation, size, model, nocs) like the reference; inputs are not modified.

    In order, each branch taken where its uniform draw is below its probability:
      bb  box deformation (:45-91): stretch along the object's axes (one in-plane factor for y-symmetric classes);
      rt  rigid perturbation (:95-130);
      bc  box-cage resize, mug and bowl only (:132-164);
      pc  Gaussian point noise of sigma ``pc_r`` (:166-169);
      nl  non-linear stretch along one axis, classes 0, 1, 2, 3, 5 (:173-215; axis 0 for the camera class 2, else 1).

    ``draws``: the random numbers, as a dict -- "prop" (B, 5) uniforms deciding the branches (bb, rt, bc, pc, nl), "bc"
    (B, 2) and "nl" (B, 2) the raw uniforms of those two branches, "noise" (B, N, 3) standard normals.  None: drawn
    here from ``generator``.  With the reference's own draws the outputs equal the reference's to float32 round-off
    (tests/test_preprocess.py against tests/golden/data_augment.npz)."""
    if not isinstance(probs, (tuple, list)):
        probs = (probs.aug_bb_pro, probs.aug_rt_pro, probs.aug_bc_pro, probs.aug_pc_pro, probs.aug_nl_pro)
    f32 = torch.float32
    dev = pts.device
    pts, model, nocs = pts.to(f32), model.to(f32), nocs.to(f32)
    rot, trans, size = rotation.to(f32), translation.to(f32).reshape(-1, 3), size.to(f32).reshape(-1, 3)
    b, n, _ = pts.shape
    obj = torch.as_tensor(obj_id, device=dev).reshape(-1).to(torch.int64)
    if draws is None:
        draws = {"prop": _draw(torch.rand, (b, 5), f32, dev, generator), "bc": _draw(torch.rand, (b, 2), f32, dev, generator),
                 "nl": _draw(torch.rand, (b, 2), f32, dev, generator), "noise": _draw(torch.randn, (b, n, 3), f32, dev, generator)}
    prop = draws["prop"].to(device=dev, dtype=f32)
    take = prop < _const([float(p) for p in probs], f32, dev)
    isin = lambda ids: (obj.unsqueeze(1) == _const(ids, torch.int64, dev)).any(1)
    do_bb, do_rt, do_pc = take[:, 0], take[:, 1], take[:, 3]
    do_bc = take[:, 2] & isin([5, 1])
    do_nl = take[:, 4] & isin([0, 1, 2, 3, 5])
    sel3 = lambda m, a, c: torch.where(m.view(-1, 1), a, c)              # (B, 3) tensors
    sel = lambda m, a, c: torch.where(m.view(-1, 1, 1), a, c)            # (B, *, 3) tensors
    norm = lambda v: torch.linalg.vector_norm(v, dim=1)

    def extents(mp):
        """(lx, ly, lz) of the deformed model points (:147-149, :195-197): symmetric extent in x, plain extents in y, z."""
        mx, mn = mp.max(dim=1).values, mp.min(dim=1).values
        return torch.stack([2 * torch.maximum(mx[:, 0], -mn[:, 0]), mx[:, 1] - mn[:, 1], mx[:, 2] - mn[:, 2]], 1)

    # ---- bb (:45-91) ----
    e = aug_bb.to(device=dev, dtype=f32).reshape(-1, 3)
    exz = (e[:, 0] + e[:, 2]) / 2
    e = sel3(torch.as_tensor(sym, device=dev).reshape(b, -1)[:, 0] == 1, torch.stack([exz, e[:, 1], exz], 1), e)
    reproj = (pts - trans.unsqueeze(1)) @ rot                            # rows: R^T (p - t)
    s_bb = size * e
    k = (norm(s_bb) / norm(size)).view(-1, 1, 1)
    pts = sel(do_bb, (reproj * e.unsqueeze(1)) @ rot.transpose(1, 2) + trans.unsqueeze(1), pts)
    nocs = sel(do_bb, nocs * e.unsqueeze(1) / k, nocs)
    model = sel(do_bb, model * e.unsqueeze(1) / k, model)
    size = sel3(do_bb, s_bb, size)

    # ---- rt (:95-130) ----
    d = aug_rt_t.to(device=dev, dtype=f32).reshape(-1, 3)
    rm = aug_rt_r.to(device=dev, dtype=f32).reshape(-1, 3, 3)
    pts = sel(do_rt, (pts + d.unsqueeze(1)) @ rm.transpose(1, 2), pts)
    trans_rt = (rm @ (trans + d).unsqueeze(2)).squeeze(2)
    rot = sel(do_rt, rm @ rot, rot)
    trans = sel3(do_rt, trans_rt, trans)

    # ---- bc (:132-164): resize x, z linearly along y ----
    ub = draws["bc"].to(device=dev, dtype=f32)
    up, down = ub[:, 0:1] * (1.2 - 0.8) + 0.8, ub[:, 1:2] * (1.2 - 0.8) + 0.8
    reproj = (pts - trans.unsqueeze(1)) @ rot
    sy = size[:, 1:2]
    xz = _const([True, False, True], torch.bool, dev)
    scale_by = lambda p, r: torch.where(xz, p * r.unsqueeze(2), p)
    pts_bc = scale_by(reproj, (reproj[:, :, 1] + sy / 2) / sy * (up - down) + down) @ rot.transpose(1, 2) + trans.unsqueeze(1)
    ns = size / norm(size).view(-1, 1)
    nsy = ns[:, 1:2]
    model_bc = scale_by(model, (model[:, :, 1] + nsy / 2) / nsy * (up - down) + down)
    ext = extents(model_bc)
    aug = norm(ext).view(-1, 1, 1)
    nocs_bc = scale_by(nocs, (nocs[:, :, 1] + nsy / 2) / nsy * (up - down) + down) / aug
    pts, model, nocs = sel(do_bc, pts_bc, pts), sel(do_bc, model_bc / aug, model), sel(do_bc, nocs_bc, nocs)
    size = sel3(do_bc, ext * norm(size).view(-1, 1), size)

    # ---- pc (:166-169) ----
    pts = sel(do_pc, pts + draws["noise"].to(device=dev, dtype=f32) * float(pc_r), pts)

    # ---- nl (:173-215): stretch one axis by a factor growing with the square of the coordinate ----
    un = draws["nl"].to(device=dev, dtype=f32)
    r_max, r_min = un[:, 0:1] * 0.2 + 1.1, -un[:, 1:2] * 0.2 + 0.9
    axis = torch.where(obj == 2, 0, 1)                                   # (:268-271)
    amask = torch.nn.functional.one_hot(axis, 3).to(torch.bool).unsqueeze(1)           # (B, 1, 3)
    pick = lambda p: torch.gather(p, 2, axis.view(-1, 1, 1).expand(-1, p.shape[1], 1)).squeeze(2)
    stretch = lambda p, r: torch.where(amask, p * r.unsqueeze(2), p)
    reproj = (pts - trans.unsqueeze(1)) @ rot
    sa = torch.gather(size, 1, axis.view(-1, 1))
    xa = pick(reproj)
    pts_nl = stretch(reproj, r_min + 4 * (xa * xa) / (sa ** 2) * (r_max - r_min)) @ rot.transpose(1, 2) + trans.unsqueeze(1)
    ns = size / norm(size).view(-1, 1)
    nsa = torch.gather(ns, 1, axis.view(-1, 1))
    ma = pick(model)
    model_nl = stretch(model, r_min + 4 * (ma * ma) / (nsa ** 2) * (r_max - r_min))
    ext = extents(model_nl)
    aug = norm(ext).view(-1, 1, 1)
    na = pick(nocs)
    nocs_nl = stretch(nocs, r_min + 4 * (na * na) / (nsa ** 2) * (r_max - r_min)) / aug
    pts, model, nocs = sel(do_nl, pts_nl, pts), sel(do_nl, model_nl / aug, model), sel(do_nl, nocs_nl, nocs)
    size = sel3(do_nl, ext * norm(size).view(-1, 1), size)
    return pts, rot, trans, size, model, nocs


def jitter_points(pts, noise=None, generator=None):
    """provider/dataset.py:211 for a batch: ``pts + clip(0.001 * N(0, 1), -0.005, 0.005)`` -- one millimetre of Gaussian
    sensor noise per coordinate, clipped at five.  The reference adds float64 noise to the float32 points and rounds to
    float32 once (``torch.FloatTensor(pts)``, :225); so does this.  ``noise``: the standard normals, same shape as ``pts``
    (None: drawn from ``generator``)."""
    if noise is None:
        noise = _draw(torch.randn, pts.shape, torch.float64, pts.device, generator)
    return (pts.to(torch.float64) + (0.001 * noise.to(device=pts.device, dtype=torch.float64)).clamp(-0.005, 0.005)).to(torch.float32)
