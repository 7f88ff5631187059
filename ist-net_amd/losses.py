"""Training losses of IST-Net (plain torch; tiny elementwise work, not part of the kernel path).

Mirror of model/losses.py:3-49 and the `SupervisedLoss` of model/ist_net.py:78-111:
loss = PoseDis(main) + PoseDis(cam aux) [+ PoseDis(world aux)] + gamma1 * SmoothL1Dis(qo) + gamma2 * MSE(features).
"""
import torch
import torch.nn as nn


def SmoothL1Dis(p1, p2, threshold=0.1):
    """p1, p2 (B,N,3): smooth-L1 per coordinate, summed over xyz, mean over points and batch."""
    from . import heads_native
    if heads_native.usable(p1, p2) and p1.shape == p2.shape and p1.shape[-1] == 3 and not p2.requires_grad:
        return heads_native.SmoothL1Function.apply(p1, p2, threshold)      # one launch per direction
    diff = torch.abs(p1 - p2)
    dis = torch.where(diff > threshold, diff - threshold / 2.0, diff.pow(2) / (2.0 * threshold))
    return torch.mean(torch.sum(dis, dim=2 if p1.dim() == 3 else 1))


def ChamferDis(p1, p2):
    """Symmetric Chamfer distance of (B,n1,3) and (B,n2,3): mean nearest-neighbour L2 both ways, halved, batch mean.
    [ref losses.py:25-34; imported but not used by the reference's IST-Net loss]"""
    pair = torch.cdist(p1, p2)                       # (B, n1, n2)
    return (0.5 * pair.min(dim=2).values.mean(dim=1) + 0.5 * pair.min(dim=1).values.mean(dim=1)).mean()


def PoseDis(r1, t1, s1, r2, t2, s2):
    """Mean column norm of R1-R2 (dim=1 as in the reference) + mean L2 of t and s differences."""
    from . import heads_native
    if (heads_native.usable(r1, t1, s1, r2, t2, s2) and r1.shape == r2.shape and r1.dim() == 3
            and not (r2.requires_grad or t2.requires_grad or s2.requires_grad)):
        return heads_native.PoseDisFunction.apply(r1, t1, s1, r2, t2, s2)   # one launch per direction
    return (torch.mean(torch.norm(r1 - r2, dim=1)) + torch.mean(torch.norm(t1 - t2, dim=1))
            + torch.mean(torch.norm(s1 - s2, dim=1)))


class _WeightedSumFn(torch.autograd.Function):
    """sum_i w_i s_i of scalar terms as one node: stack + dot forward, one scaled copy of the weights backward -- the
    reference's chain ``loss = a + b + g1 * c + g2 * d + e`` (model/ist_net.py:95-110) is four adds and two multiplies
    forward and two multiplies backward, each a launch at the turn from forward to backward."""

    @staticmethod
    def forward(ctx, weights, *terms):
        ctx.save_for_backward(weights)
        return torch.dot(torch.stack([t.reshape(()) for t in terms]), weights)

    @staticmethod
    def backward(ctx, g):
        (weights,) = ctx.saved_tensors
        return (None, *(g * weights).unbind(0))


class _MseFn(torch.autograd.Function):
    """nn.functional.mse_loss(a, b) with value and gradient from ONE pass over the operands (mse_value_and_grad)."""

    @staticmethod
    def forward(ctx, a, b):
        loss, g = mse_value_and_grad(a, b)
        ctx.save_for_backward(g)
        return loss

    @staticmethod
    def backward(ctx, gl):
        (g,) = ctx.saved_tensors
        ga = g * gl
        return (ga if ctx.needs_input_grad[0] else None, -ga if ctx.needs_input_grad[1] else None)


def _mse(a, b):
    if (a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.shape == b.shape and a.is_contiguous()
            and b.is_contiguous() and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0):
        return _MseFn.apply(a, b)
    return nn.functional.mse_loss(a, b)


class SupervisedLoss(nn.Module):
    def __init__(self, gamma1=1.0, gamma2=10.0, freeze_world_enhancer=False):
        super().__init__()
        self.gamma1, self.gamma2, self.freeze_world_enhancer = gamma1, gamma2, freeze_world_enhancer
        self._weights = {}

    def forward(self, end_points):
        ep = end_points
        labels = (ep["rotation_label"], ep["translation_label"], ep["size_label"])
        terms = [PoseDis(ep["pred_rotation"], ep["pred_translation"], ep["pred_size"], *labels),
                 PoseDis(ep["pred_rotation_aux_cam"], ep["pred_translation_aux_cam"], ep["pred_size_aux_cam"], *labels),
                 SmoothL1Dis(ep["pred_qo"], ep["qo"]),
                 _mse(ep["pts_w_local"], ep["pts_w_local_gt"])]
        weights = [1.0, 1.0, self.gamma1, self.gamma2]
        if not self.freeze_world_enhancer:
            terms.append(PoseDis(ep["pred_rotation_aux_world"], ep["pred_translation_aux_world"],
                                 ep["pred_size_aux_world"], *labels))
            weights.append(1.0)
        if not terms[0].is_cuda:
            loss = terms[0]                          # host: the reference's chain as it is written
            for w, t in zip(weights[1:], terms[1:]):
                loss = loss + (t if w == 1.0 else w * t)
            return loss
        key = (terms[0].device, terms[0].dtype, tuple(weights))
        if key not in self._weights:
            self._weights[key] = torch.tensor(weights, dtype=terms[0].dtype, device=terms[0].device)
        return _WeightedSumFn.apply(self._weights[key], *terms)


def mse_value_and_grad(a, b=None):
    """``nn.functional.mse_loss(a, b)`` (``b=None``: against zero, i.e. the mean of squares) together with its gradient with
    respect to ``a`` -- (loss, 2 (a - b) / n) -- from ONE pass over the operands (include/istnet_heads.h,
    istnet_mse_value_grad) for CUDA float32 tensors; the framework's mse_loss + autograd are five launches over the same
    bytes.  For loops that start backward from an explicit gradient: ``loss, g = mse_value_and_grad(out); out.backward(g)``
    (what bench.py does for its encoder workload); the gradient with respect to ``b`` is ``-g``.  No graph is recorded."""
    x = a.detach()
    y = b.detach() if b is not None else None
    if (not x.is_cuda or x.dtype != torch.float32 or not x.is_contiguous() or x.data_ptr() % 16
            or (y is not None and (y.shape != x.shape or y.dtype != x.dtype or not y.is_contiguous() or y.data_ptr() % 16))):
        # (the native pass reads float4: a contiguous view at an odd storage offset takes the framework's ops too)
        d = x if y is None else x - y
        return d.square().mean(), d * (2.0 / d.numel())
    from . import _native
    lib = _native.lib()
    n = x.numel()
    g = torch.empty_like(x)
    part = torch.empty(lib.istnet_mse_parts(n), dtype=torch.float32, device=x.device)
    loss = torch.empty((), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _native.check(lib.istnet_mse_value_grad(n, x.data_ptr(), y.data_ptr() if y is not None else None, g.data_ptr(),
                                                part.data_ptr(), loss.data_ptr(),
                                                torch.cuda.current_stream(x.device).cuda_stream), "mse_value_grad")
    return loss, g
