"""Native tail of the pose estimators and of the pose / NOCS losses (csrc/pose_tail.hip, include/istnet_heads.h).

Reference: model/ist_net.py:250-264,318-332 (three heads Linear 512-512-256-k + ReLU on the mean-pooled feature, then
Ortho6d2Mat), utils/rotation_utils.py:4-28, model/losses.py:3-49.  Each piece is ONE autograd node with one launch per
direction (three per direction for the three-layer heads) instead of 10-45 framework ops each: a training step of the
full model went from ~1 100 framework launches to a few hundred (profiles/r03_heads_native.txt).

CUDA float32 only; CPU tensors (host-logic tests, cpu_baseline) keep the literal torch composition in the callers.
"""
import ctypes

import torch
from torch.autograd import Function

from . import _native

USE_NATIVE_TAIL = True      # tests flip it to compare against the torch composition


def usable(*tensors):
    return USE_NATIVE_TAIL and all(t.is_cuda and t.dtype == torch.float32 for t in tensors)


def _st(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _ptrs(tensors):
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def _grad_dest(param, shape):
    from .pointnet2.fused_mlp import _grad_dest as dest
    return dest(param, tuple(shape), param.device)


class Ortho6dFunction(Function):
    """r6 (B, 6) = [x_raw | y_raw] -> R (B, 3, 3)  (utils/rotation_utils.py:21-28)."""

    @staticmethod
    def forward(ctx, r6):
        r6 = r6.contiguous()
        b = r6.shape[0]
        out = torch.empty((b, 3, 3), dtype=torch.float32, device=r6.device)
        with torch.cuda.device(r6.device):
            _native.check(_native.lib().istnet_ortho6d_forward(b, r6.data_ptr(), out.data_ptr(), _st(r6.device)),
                          "ortho6d_forward")
        ctx.save_for_backward(r6)
        return out

    @staticmethod
    def backward(ctx, d_r):
        (r6,) = ctx.saved_tensors
        d_r = d_r.contiguous()
        d_r6 = torch.empty_like(r6)
        with torch.cuda.device(r6.device):
            _native.check(_native.lib().istnet_ortho6d_backward(r6.shape[0], r6.data_ptr(), d_r.data_ptr(), d_r6.data_ptr(),
                                                                _st(r6.device)), "ortho6d_backward")
        return d_r6


class PoseDisFunction(Function):
    """PoseDis(r1, t1, s1, r2, t2, s2) of model/losses.py:37-49 as one scalar; gradients for the first triple only."""

    @staticmethod
    def forward(ctx, r1, t1, s1, r2, t2, s2):
        ts = [t.contiguous() for t in (r1, t1, s1, r2, t2, s2)]
        b, dev = ts[0].shape[0], ts[0].device
        loss = torch.empty((), dtype=torch.float32, device=dev)
        norms = torch.empty((b, 5), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _native.check(_native.lib().istnet_pose_dis_forward(b, *[t.data_ptr() for t in ts], loss.data_ptr(),
                                                                norms.data_ptr(), _st(dev)), "pose_dis_forward")
        ctx.save_for_backward(*ts, norms)
        return loss

    @staticmethod
    def backward(ctx, gout):
        r1, t1, s1, r2, t2, s2, norms = ctx.saved_tensors
        dev, b = r1.device, r1.shape[0]
        gout = gout.contiguous()
        dr, dt, ds = torch.empty_like(r1), torch.empty_like(t1), torch.empty_like(s1)
        with torch.cuda.device(dev):
            _native.check(_native.lib().istnet_pose_dis_backward(
                b, gout.data_ptr(), r1.data_ptr(), t1.data_ptr(), s1.data_ptr(), r2.data_ptr(), t2.data_ptr(), s2.data_ptr(),
                norms.data_ptr(), dr.data_ptr(), dt.data_ptr(), ds.data_ptr(), _st(dev)), "pose_dis_backward")
        return dr, dt, ds, None, None, None


class SmoothL1Function(Function):
    """SmoothL1Dis(p1, p2) of model/losses.py:3-22 on (..., 3) tensors; gradient for p1 only."""

    @staticmethod
    def forward(ctx, p1, p2, threshold):
        p1, p2 = p1.contiguous(), p2.contiguous()
        dev, rows = p1.device, p1.numel() // 3
        lib = _native.lib()
        part = torch.empty((lib.istnet_smooth_l1_parts(rows),), dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _native.check(lib.istnet_smooth_l1_forward(rows, float(threshold), p1.data_ptr(), p2.data_ptr(), part.data_ptr(),
                                                       loss.data_ptr(), _st(dev)), "smooth_l1_forward")
        ctx.save_for_backward(p1, p2)
        ctx.threshold = float(threshold)
        return loss

    @staticmethod
    def backward(ctx, gout):
        p1, p2 = ctx.saved_tensors
        dev = p1.device
        dp1 = torch.empty_like(p1)
        with torch.cuda.device(dev):
            _native.check(_native.lib().istnet_smooth_l1_backward(p1.numel() // 3, ctx.threshold, gout.contiguous().data_ptr(),
                                                                  p1.data_ptr(), p2.data_ptr(), dp1.data_ptr(), _st(dev)),
                          "smooth_l1_backward")
        return dp1, None, None


class FCHeadsFunction(Function):
    """The three-layer heads of one estimator side by side: pooled (B, K) and, per head h, (w1, b1, w2, b2, w3, b3) ->
    one output (B, n3[h]) per head.  Layers 1 and 2 are followed by ReLU (model/ist_net.py:232-248).  Three launches forward;
    backward two launches per layer (input gradients, weight / bias gradients), the heads' gradients with respect to the
    shared pooled feature summed inside the layer-1 launch."""

    @staticmethod
    def forward(ctx, pooled, nheads, *params):
        lib = _native.lib()
        dev = pooled.device
        pooled = pooled.contiguous()
        b = pooled.shape[0]
        heads = [params[6 * h:6 * h + 6] for h in range(nheads)]
        cur = [pooled] * nheads
        acts = []
        with torch.cuda.device(dev):
            st = _st(dev)
            for layer in range(3):
                ws = [hd[2 * layer] for hd in heads]
                bs = [hd[2 * layer + 1] for hd in heads]
                k = ws[0].shape[1]
                n = [w.shape[0] for w in ws]
                ys = [torch.empty((b, nn_), dtype=torch.float32, device=dev) for nn_ in n]
                _native.check(lib.istnet_fc_forward(nheads, b, k, (ctypes.c_int * nheads)(*n), _ptrs(cur), _ptrs(ws),
                                                    _ptrs(bs), _ptrs(ys), 1 if layer < 2 else 0, st), "fc_forward")
                acts.append(ys)
                cur = ys
        ctx.nheads = nheads
        ctx.save_for_backward(pooled, *acts[0], *acts[1], *params)
        return tuple(acts[2])

    @staticmethod
    def backward(ctx, *d_out):
        lib = _native.lib()
        nh = ctx.nheads
        sv = ctx.saved_tensors
        pooled, y1, y2, params = sv[0], sv[1:1 + nh], sv[1 + nh:1 + 2 * nh], sv[1 + 2 * nh:]
        heads = [params[6 * h:6 * h + 6] for h in range(nh)]
        dev, b = pooled.device, pooled.shape[0]
        grads = [None] * (6 * nh)
        need_pooled = ctx.needs_input_grad[0]
        d_cur = [(g if g is not None else torch.zeros((b, heads[h][4].shape[0]), device=dev)).contiguous()
                 for h, g in enumerate(d_out)]
        with torch.cuda.device(dev):
            st = _st(dev)
            for layer in (2, 1, 0):
                ws = [hd[2 * layer] for hd in heads]
                bs = [hd[2 * layer + 1] for hd in heads]
                k = ws[0].shape[1]
                n = [w.shape[0] for w in ws]
                xs = [pooled] * nh if layer == 0 else (y1 if layer == 1 else y2)
                ys = [None] * nh if layer == 2 else (y2 if layer == 1 else y1)      # this layer's output: the relu mask
                relu = 0 if layer == 2 else 1
                shared = 1 if layer == 0 else 0
                if layer > 0:
                    dxs = [torch.empty((b, k), dtype=torch.float32, device=dev) for _ in range(nh)]
                elif need_pooled:
                    dxs = [torch.empty((b, k), dtype=torch.float32, device=dev)] + [None] * (nh - 1)
                else:
                    dxs = None
                dws = [_grad_dest(w, w.shape) for w in ws]
                dbs = [_grad_dest(bb, bb.shape) for bb in bs]
                _native.check(lib.istnet_fc_backward(
                    nh, b, k, (ctypes.c_int * nh)(*n), _ptrs(d_cur), _ptrs(ys) if relu else None, _ptrs(ws), _ptrs(xs),
                    _ptrs(dxs) if dxs is not None else None, _ptrs(dws), _ptrs(dbs), relu, shared, st), "fc_backward")
                for h in range(nh):
                    grads[6 * h + 2 * layer] = dws[h].view_as(ws[h])
                    grads[6 * h + 2 * layer + 1] = dbs[h].view_as(bs[h])
                d_cur = dxs
        d_pooled = d_cur[0] if (need_pooled and d_cur is not None) else None
        return (d_pooled, None, *grads)


def fc_heads(heads, pooled):
    """``[head(pooled) for head in heads]`` for ``nn.Sequential(Linear, ReLU, Linear, ReLU, Linear)`` heads on a CUDA
    float32 (B <= 64, K % 4 == 0) feature; None when the fused form does not apply (the caller runs the modules)."""
    if not usable(pooled) or pooled.dim() != 2 or pooled.shape[0] > 64 or not 1 <= len(heads) <= 4:
        return None
    params = []
    for head in heads:
        mods = list(head)
        if not (len(mods) == 5 and all(isinstance(mods[i], torch.nn.Linear) and mods[i].bias is not None for i in (0, 2, 4))
                and all(isinstance(mods[i], torch.nn.ReLU) for i in (1, 3))):
            return None
        if mods[0].in_features != pooled.shape[1] or any(mods[i].in_features % 4 for i in (0, 2, 4)):
            return None
        for i in (0, 2, 4):
            params += [mods[i].weight, mods[i].bias]
    ws = params[0::2]          # [head][layer] weights, layer-minor
    if not all(ws[l].shape[1] == ws[3 * h + l].shape[1] for h in range(len(heads)) for l in range(3)):
        return None        # the heads of one launch share K per layer
    return FCHeadsFunction.apply(pooled, len(heads), *params)
