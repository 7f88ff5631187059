"""Fused SharedMLP (+ max over nsample) on the gfx950 MFMA kernels of csrc/pw_mlp.hip.

Replaces, for CUDA tensors, the reference's per-scale tail of a set-abstraction level
(pointnet2_modules.py:61-68: ``mlps[i](grouped)`` then ``F.max_pool2d`` then ``squeeze``) and the
feature-propagation MLP (pointnet2_modules.py:205-209) with one autograd node:

    forward : per layer  y_l = W_l . relu(bn_{l-1}(y_{l-1}))  (+ BN statistics),  then
              out = max_s relu(bn_L(y_L))        -- only raw pre-BN activations are stored
    backward: per layer  BN/ReLU/max-pool gradients are folded into the operand staging of the
              dgrad and wgrad GEMMs (include/istnet_pw.h)

Numerics: exact-f32 MFMA (no TF32 / bf16); BatchNorm statistics are reduced in float64 from
per-tile f32 partials; results match torch's Conv2d/BatchNorm2d/ReLU/max_pool2d to ~1e-6 relative
(the 1e-4 bar of BASELINE.json).  Training-mode running statistics follow torch (momentum,
unbiased variance); ``num_batches_tracked`` is incremented by the caller.
"""
import os

import torch
from torch.autograd import Function

from .. import _native


if os.environ.get("ISTNET_POISON_ALLOC"):  # debugging aid: expose reads of never-written workspace
    def _empty(shape, dtype, device):
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        if dtype == torch.uint8:
            return torch.full(shape, 255, dtype=dtype, device=device)
        return torch.full(shape, float("nan"), dtype=dtype, device=device)
else:
    def _empty(shape, dtype, device):
        return torch.empty(shape, dtype=dtype, device=device)


def _kname(base, cfg):
    """Kernel symbol as rocprofv3 prints it, e.g. pw_dgrad_kernel<128, 128, 2, 2>."""
    mt, nt = cfg // 1000, cfg % 1000
    wm, wn = (1, 4) if mt == 32 else (2, 2)
    return f"{base}<{mt}, {nt}, {wm}, {wn}>"


def _st(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


class _Layer:
    """Per-layer constants handed to the autograd function (not differentiable)."""
    __slots__ = ("running_mean", "running_var", "momentum", "eps")

    def __init__(self, bn):
        self.running_mean = bn.running_mean
        self.running_var = bn.running_var
        self.momentum = bn.momentum
        self.eps = bn.eps


class FusedSharedMLPFunction(Function):
    """x (B, C0, G, S) -> (B, C_L, G): conv1x1/BN/ReLU stack followed by a max over S."""

    @staticmethod
    def forward(ctx, x, training, layers, *params):
        lib = _native.lib()
        dev = x.device
        b, c0, g, s = x.shape
        p = g * s
        st = _st(dev)
        n_layers = len(layers)
        x = x.contiguous()
        ys, bns, wts = [], [], []
        cur, cur_c, in_bn = x, c0, None
        with torch.cuda.device(dev):
            for li, lay in enumerate(layers):
                w, gamma, beta = params[3 * li], params[3 * li + 1], params[3 * li + 2]
                cout = w.shape[0]
                w2 = w.reshape(cout, cur_c)
                wt = w2.t().contiguous()
                y = _empty((b, cout, p), torch.float32, dev)
                bn = _empty((4, cout), torch.float32, dev)
                if training:
                    nt = lib.istnet_pw_stat_tiles(b, cout, p)
                    part = _empty((2, cout, nt), torch.float32, dev)
                    ps, pq = part[0].data_ptr(), part[1].data_ptr()
                else:
                    nt, ps, pq = 0, None, None
                sc, sh = (_p(in_bn[0]), _p(in_bn[1])) if in_bn is not None else (None, None)
                cin_l, src = cur_c, cur
                _native.check(_native.timed(
                    _kname("pw_fwd_kernel", lib.istnet_pw_tile_cfg(b, cout, p)), 2.0 * b * p * cin_l * cout,
                    4.0 * b * p * (cin_l + cout),
                    lambda: lib.istnet_pw_forward(b, cin_l, cout, p, src.data_ptr(), wt.data_ptr(), sc, sh,
                                                  y.data_ptr(), ps, pq, st)), "pw_forward")
                if training:
                    _native.check(lib.istnet_bn_finalize_fwd(
                        cout, nt, float(b * p), ps, pq, gamma.data_ptr(), beta.data_ptr(), float(lay.eps),
                        float(lay.momentum), _p(lay.running_mean), _p(lay.running_var), bn.data_ptr(), st),
                        "bn_finalize_fwd")
                else:
                    istd = torch.rsqrt(lay.running_var + lay.eps)
                    bn[0] = gamma * istd
                    bn[1] = beta - lay.running_mean * bn[0]
                    bn[2] = lay.running_mean
                    bn[3] = istd
                ys.append(y)
                bns.append(bn)
                wts.append(w2)
                cur, cur_c, in_bn = y, cout, bn
            out = _empty((b, cur_c, g), torch.float32, dev)
            arg = _empty((b, cur_c, g), torch.uint8, dev) if s > 1 else None
            _native.check(lib.istnet_bn_relu_pool(b, cur_c, g, s, cur.data_ptr(), in_bn.data_ptr(),
                                                  out.data_ptr(), _p(arg), st), "bn_relu_pool")
        ctx.training = training
        ctx.shape = (b, c0, g, s)
        ctx.n_layers = n_layers
        ctx.save_for_backward(x, arg if arg is not None else torch.empty(0, device=dev), *ys, *bns, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _native.lib()
        b, c0, g, s = ctx.shape
        p = g * s
        n = ctx.n_layers
        saved = ctx.saved_tensors
        x, arg = saved[0], saved[1]
        ys, bns, params = saved[2:2 + n], saved[2 + n:2 + 2 * n], saved[2 + 2 * n:]
        dev = x.device
        st = _st(dev)
        dout = dout.contiguous()
        grads = [None] * (3 * n)
        pooled = s > 1
        d_dense, d_pooled, d_arg = (None, dout, arg) if pooled else (dout, None, None)
        dx = None
        with torch.cuda.device(dev):
            ntb = lib.istnet_pw_bwd_stat_tiles(b, p)
            for li in range(n - 1, -1, -1):
                w, gamma = params[3 * li], params[3 * li + 1]
                cout = w.shape[0]
                cin = c0 if li == 0 else params[3 * (li - 1)].shape[0]
                w2 = w.reshape(cout, cin)
                y, bn = ys[li], bns[li]
                part = _empty((2, cout, ntb), torch.float32, dev)
                _native.check(lib.istnet_pw_bwd_stats(
                    b, cout, p, s if pooled and li == n - 1 else 0, y.data_ptr(), _p(d_dense), _p(d_pooled),
                    _p(d_arg), bn.data_ptr(), part[0].data_ptr(), part[1].data_ptr(), st), "pw_bwd_stats")
                dgamma = _empty(cout, torch.float32, dev)
                dbeta = _empty(cout, torch.float32, dev)
                bwdc = _empty((3, cout), torch.float32, dev)
                _native.check(lib.istnet_bn_finalize_bwd(
                    cout, ntb, float(b * p), 1 if ctx.training else 0, part[0].data_ptr(), part[1].data_ptr(),
                    gamma.data_ptr(), bn.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), bwdc.data_ptr(), st),
                    "bn_finalize_bwd")
                src = x if li == 0 else ys[li - 1]
                in_bn = None if li == 0 else bns[li - 1]
                ns_arg = s if pooled and li == n - 1 else 0
                if ctx.needs_input_grad[3 + 3 * li]:
                    splits = lib.istnet_pw_wgrad_splits(b, cin, cout, p)
                    ws = _empty((splits, cout, cin), torch.float32, dev)
                    dw = _empty((cout, cin), torch.float32, dev)
                    sc, sh = (_p(in_bn[0]), _p(in_bn[1])) if in_bn is not None else (None, None)
                    dd, dp, da = _p(d_dense), _p(d_pooled), _p(d_arg)
                    grad_elems = b * cout * (p if dd is not None else p // s)
                    _native.check(_native.timed(
                        _kname("pw_wgrad_kernel", lib.istnet_pw_wgrad_tile_cfg(cin, cout)), 2.0 * b * p * cin * cout,
                        4.0 * (b * p * (cin + cout) + grad_elems),
                        lambda: lib.istnet_pw_wgrad(b, cin, cout, p, ns_arg, src.data_ptr(), sc, sh, y.data_ptr(),
                                                    dd, dp, da, bn.data_ptr(), bwdc.data_ptr(), ws.data_ptr(),
                                                    st)), "pw_wgrad")
                    _native.check(lib.istnet_pw_wgrad_reduce(cout * cin, splits, ws.data_ptr(), dw.data_ptr(), st),
                                  "pw_wgrad_reduce")
                    grads[3 * li] = dw.view_as(w)
                grads[3 * li + 1] = dgamma
                grads[3 * li + 2] = dbeta
                if li > 0 or ctx.needs_input_grad[0]:
                    dprev = _empty((b, cin, p), torch.float32, dev)
                    dd, dp, da = _p(d_dense), _p(d_pooled), _p(d_arg)
                    grad_elems = b * cout * (p if dd is not None else p // s)
                    _native.check(_native.timed(
                        _kname("pw_dgrad_kernel", lib.istnet_pw_tile_cfg(b, cin, p)), 2.0 * b * p * cin * cout,
                        4.0 * (b * p * (cin + cout) + grad_elems),
                        lambda: lib.istnet_pw_dgrad(b, cin, 0, cin, cout, p, ns_arg, w2.data_ptr(), y.data_ptr(),
                                                    dd, dp, da, bn.data_ptr(), bwdc.data_ptr(), dprev.data_ptr(),
                                                    st)), "pw_dgrad")
                    d_dense, d_pooled, d_arg = dprev, None, None
                    if li == 0:
                        dx = dprev.view(b, c0, g, s)
        return (dx, None, None, *grads)


def _fusable(mlp, x):
    """True when `mlp` is a plain [conv1x1(no bias) -> BatchNorm2d -> ReLU] x k stack on a CUDA f32 tensor."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        return False
    s = x.shape[3]
    if (x.shape[2] * s) % 4 != 0 or (s > 1 and s not in (4, 8, 16, 32, 64)):
        return False
    for unit in mlp:
        names = [n for n, _ in unit.named_children()]
        if names != ["conv", "normlayer", "activation"]:
            return False
        conv, bn, act = unit.conv, unit.normlayer.bn, unit.activation
        if not (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (1, 1) and conv.bias is None
                and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1):
            return False
        if not (isinstance(bn, torch.nn.BatchNorm2d) and bn.affine and bn.track_running_stats
                and bn.momentum is not None):
            return False
        if not isinstance(act, torch.nn.ReLU):
            return False
    return len(mlp) > 0


def shared_mlp_maxpool(mlp, x):
    """``max_pool2d(mlp(x), [1, nsample]).squeeze(-1)`` for x (B, C, npoint, nsample) -> (B, C', npoint).

    CUDA tensors of a supported shape run the fused HIP path; anything else (CPU tensors in the
    host-logic tests, exotic module stacks) takes the reference composition with torch ops.
    """
    if not _fusable(mlp, x):
        act = mlp(x)
        return torch.nn.functional.max_pool2d(act, kernel_size=[1, act.size(3)]).squeeze(-1)
    layers, params = [], []
    for unit in mlp:
        bn = unit.normlayer.bn
        layers.append(_Layer(bn))
        params += [unit.conv.weight, bn.weight, bn.bias]
    training = mlp.training
    out = FusedSharedMLPFunction.apply(x, training, layers, *params)
    if training:
        for unit in mlp:
            unit.normlayer.bn.num_batches_tracked += 1
    return out
