"""6-D rotation representation -> SO(3) by Gram-Schmidt.

Mirror of utils/rotation_utils.py:4-28, device-agnostic (the reference hard-codes ``.cuda()``
at :6).  Columns of the result are [x, y, z] with y = norm(y_raw), z = norm(x_raw x y), x = y x z;
vector norms are clamped at 1e-8.
"""
import torch


def normalize_vector(v, dim=1, return_mag=False):
    mag = torch.sqrt(v.pow(2).sum(dim=dim, keepdim=True))
    mag = torch.clamp(mag, min=1e-8)
    out = v / mag.expand_as(v)
    return (out, mag) if return_mag else out


def cross_product(u, v):
    i = u[:, 1] * v[:, 2] - u[:, 2] * v[:, 1]
    j = u[:, 2] * v[:, 0] - u[:, 0] * v[:, 2]
    k = u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]
    return torch.stack((i, j, k), dim=1)


def Ortho6d2Mat(x_raw, y_raw):
    """(B,3), (B,3) -> (B,3,3)."""
    from . import heads_native
    if heads_native.usable(x_raw, y_raw) and x_raw.dim() == 2:     # one launch per direction (csrc/pose_tail.hip)
        return heads_native.Ortho6dFunction.apply(torch.cat((x_raw, y_raw), dim=1))
    y = normalize_vector(y_raw)
    z = normalize_vector(cross_product(x_raw, y))
    x = cross_product(y, z)
    return torch.stack((x, y, z), dim=2)


def ortho6d_to_mat(r6):
    """``Ortho6d2Mat(r6[:, :3], r6[:, 3:])`` for the (B, 6) output of a rotation head, without the slices / concat."""
    from . import heads_native
    if heads_native.usable(r6) and r6.dim() == 2 and r6.shape[1] == 6:
        return heads_native.Ortho6dFunction.apply(r6)
    return Ortho6d2Mat(r6[:, :3].contiguous(), r6[:, 3:].contiguous())
