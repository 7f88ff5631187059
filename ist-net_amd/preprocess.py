"""Per-instance input preparation on the GPU.

``backproject_choose`` replaces the numpy block of the reference's Dataset classes that turns a depth image, an
instance crop and the sampled pixel list into the network inputs ``pts`` and ``choose``
(provider/dataset.py:203-210,226-231 for training, :348-355,392,401-405 for testing).  The host code there builds
the whole (480,640,3) float64 point map for every image; the kernel (csrc/preproc.hip through
include/istnet_preproc.h) back-projects only the sampled pixels, bit-exactly.  Mask logic, random sampling
(`np.random.choice`), image decoding and the RGB resize stay with the caller.
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
        img = img.view(torch.int16).to(torch.int32) & 0xffff            # raw millimetres, stored as 16-bit
    # numpy evaluates dpt / cam_scale * scale_2_80m in float64 and fill_in_multiscale rounds to float32 once
    depth = (img.to(torch.float64) / float(cam_scale) * float(scale_2_80m)).to(torch.float32).contiguous()
    b, h, w = depth.shape
    lib = _native.lib()
    scratch = torch.empty(lib.istnet_depth_fill_scratch_floats(b, h, w), dtype=torch.float32, device=depth.device)
    out = torch.empty_like(depth)
    with torch.cuda.device(depth.device):
        _native.check(lib.istnet_depth_fill_multiscale(b, h, w, depth.data_ptr(), 3.0, scratch.data_ptr(), out.data_ptr(),
                                                       torch.cuda.current_stream(depth.device).cuda_stream),
                      "depth_fill_multiscale")
    out = out / float(scale_2_80m) * float(cam_scale)
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
