"""Per-point MLP stacks of the IST head and the pose heads (reference model/ist_net.py:130-160, 206-248, 271-316) on the fused
kernels: ``FusedBiasMLPFunction`` (Conv1d + bias (+ ReLU) stacks), ``FusedMultiSourceBiasMLPFunction`` (layer 0 over several
source tensors without the concatenation, a per-cloud mean input as a rank-1 bias, the mean over N folded into the tail) and
the dispatchers ``pointwise_conv_stack`` / ``pointwise_conv_stack_multi``.  Split out of fused_mlp.py in round 5; re-exported
from there."""
import torch
from torch.autograd import Function

from .. import _native
from .fused_mlp import (_Layer, _backward_stack, _dgrad_kname, _empty, _enter_backward, _forward_stack, _grad_dest, _head_stack,
                        _kname, _wgrad_kname, _note_fallback, _p, _st)


_ONES = {}


def _ones(dev, c):
    key = (dev.index, c)
    if key not in _ONES:
        _ONES[key] = torch.ones(c, dtype=torch.float32, device=dev)
    return _ONES[key]


class FusedBiasMLPFunction(Function):
    """x (B, C0, N) -> (B, C_L, N): stack of Conv1d(k=1) + bias (+ ReLU) layers -- the per-point MLPs of the
    IST head and the pose heads (reference model/ist_net.py:130-160, 206-248, 271-316).

    Same kernels as the BatchNorm stack with constant "BN" blocks (scale 1, shift = bias): each layer is
    one MFMA GEMM whose operand loader applies the previous layer's bias + ReLU; backward uses
    dY = g, dbias = sum g.  ``relu_last`` tells whether the last conv is followed by a ReLU."""

    @staticmethod
    def forward(ctx, x, relu_last, *params):          # params = [w0, b0, w1, b1, ...]
        lib = _native.lib()
        dev = x.device
        b, c0, npts = x.shape
        x = x.contiguous()
        n = len(params) // 2
        layers = [_Layer(None, relu=(relu_last or li < n - 1)) for li in range(n)]
        ones = [_ones(dev, params[2 * li].shape[0]) for li in range(n)]   # stand-in "gamma" of a bias layer
        flat = []
        for li in range(n):
            flat += [params[2 * li], ones[li], params[2 * li + 1]]
        with torch.cuda.device(dev):
            out, _, ys, bns = _forward_stack(lib, dev, _st(dev), b, c0, npts, 1, x, None, False, layers, flat)
        ctx.shape = (b, c0, npts)
        ctx.n_layers = n
        ctx.save_for_backward(x, *ys, *bns, *flat)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _native.lib()
        b, c0, npts = ctx.shape
        n = ctx.n_layers
        saved = ctx.saved_tensors
        x = saved[0]
        ys, bns, flat = saved[1:1 + n], saved[1 + n:1 + 2 * n], saved[1 + 2 * n:]
        dev = x.device
        _enter_backward(dev)
        need_w = [ctx.needs_input_grad[2 + 2 * li] for li in range(n)]
        with torch.cuda.device(dev), _head_stack():
            grads, dx, _ = _backward_stack(lib, dev, _st(dev), b, c0, npts, 1, x, None, False, ys, bns, flat,
                                           None, dout.contiguous(), need_w, ctx.needs_input_grad[0])
        out = []
        for li in range(n):
            dw = grads[3 * li]
            out += [dw.view_as(flat[3 * li]) if dw is not None else None, grads[3 * li + 2]]   # dW, dbias
        return (dx, None, *out)


class FusedMultiSourceBiasMLPFunction(Function):
    """Per-point Conv1d(k=1) + bias (+ ReLU) stack whose input is the channel concatenation of several tensors
    (B, C_i, N) -- optionally followed by the per-cloud mean of the (single) source expanded over the points -- WITHOUT
    building that input.  The IST head and the pose heads concatenate 3-5 feature tensors, or a feature map with its
    global mean, in front of every stack (reference model/ist_net.py:167-175,253-257,322-325: up to 512 channels x
    32 768 points = 67 MB per concat, written and read back).  Here layer 0 walks the sources in its K loop
    (istnet_pw_forward_multi), and with W0 = [Wa | Wb] the mean part is the rank-1 per-cloud bias Wb . mean(feat):

        y0 = sum_i W0[:, slice_i] . src_i  (+ (Wb . mean_b)[:, None])  + bias

    Backward: d src_i = W0[:, slice_i]^T . dY0 and dW0[:, slice_i] = dY0 . src_i^T per source (existing dgrad / wgrad
    kernels on the slices); the mean term adds Wb^T . (sum_p dY0) / N to every point of the cloud and
    dWb = sum_b (sum_p dY0[b]) (x) mean_b, with the per-cloud sums of dY0 taken from the statistics partials the
    layer-1 dgrad already produced.  tensors = [src_0 .. src_{k-1}, w0, b0, w1, b1, ...]."""

    @staticmethod
    def forward(ctx, nsrc, with_mean, relu_last, pool_mean, *tensors):
        # pool_mean: return the mean over the points of the stack's (ReLU) output, (B, C_L) -- the AdaptiveAvgPool1d(1) that
        # ends pose_mlp2 (model/ist_net.py:246,314) -- from the last RAW output in one pass (istnet_bn_relu_mean): the
        # (B, C_L, N) activation is neither written nor read back
        import ctypes
        lib = _native.lib()
        srcs = [t.contiguous() for t in tensors[:nsrc]]
        params = tensors[nsrc:]
        dev = srcs[0].device
        b, _, npts = srcs[0].shape
        chans = [t.shape[1] for t in srcs]
        csum = sum(chans)
        n = len(params) // 2
        w0, b0 = params[0], params[1]
        cout0, cin_total = w0.shape[0], w0.shape[1]
        w2 = w0.reshape(cout0, cin_total)
        layers = [_Layer(None, relu=(relu_last or li < n - 1)) for li in range(n)]
        ones = [_ones(dev, params[2 * li].shape[0]) for li in range(n)]
        flat = []
        for li in range(n):
            flat += [params[2 * li], ones[li], params[2 * li + 1]]
        with torch.cuda.device(dev):
            st = _st(dev)
            mean = row_init = None
            if with_mean:
                mean = srcs[0].mean(dim=2)                               # (B, C)
                row_init = torch.matmul(mean, w2[:, csum:].t()).contiguous()   # (B, cout0) = (Wb . mean_b)^T
            y0 = _empty((b, cout0, npts), torch.float32, dev)
            _native.check(lib.istnet_pw_forward_multi(
                b, nsrc, (ctypes.c_void_p * nsrc)(*[t.data_ptr() for t in srcs]), (ctypes.c_int * nsrc)(*chans), cout0,
                npts, w2.data_ptr(), cin_total, _p(row_init), y0.data_ptr(), st), "pw_forward_multi")
            bn0 = _empty((4, cout0), torch.float32, dev)
            _native.check(lib.istnet_affine_consts(cout0, None, b0.data_ptr(), None, None, 0.0, bn0.data_ptr(), st),
                          "affine_consts")
            out, _, ys, bns = _forward_stack(lib, dev, st, b, cin_total, npts, 1, None, None, False, layers, flat,
                                             start=(y0, bn0), tail=not pool_mean)
            if pool_mean:
                c_last = flat[-3].shape[0]
                out = _empty((b, c_last), torch.float32, dev)
                _native.check(lib.istnet_bn_relu_mean(b, c_last, npts, ys[-1].data_ptr(), bns[-1].data_ptr(), out.data_ptr(),
                                                      st), "bn_relu_mean")
        ctx.meta = (nsrc, with_mean, b, npts, chans, n)
        ctx.pool_mean = pool_mean
        ctx.save_for_backward(*srcs, mean if mean is not None else torch.empty(0, device=dev), *ys, *bns, *flat)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _native.lib()
        nsrc, with_mean, b, npts, chans, n = ctx.meta
        sv = ctx.saved_tensors
        srcs, mean = sv[:nsrc], sv[nsrc]
        ys, bns, flat = sv[nsrc + 1:nsrc + 1 + n], sv[nsrc + 1 + n:nsrc + 1 + 2 * n], sv[nsrc + 1 + 2 * n:]
        dev = srcs[0].device
        _enter_backward(dev)
        w0 = flat[0]
        cout0, cin_total = w0.shape[0], w0.shape[1]
        w2 = w0.reshape(cout0, cin_total)
        csum = sum(chans)
        need_src = [ctx.needs_input_grad[4 + i] for i in range(nsrc)]
        need_w = [ctx.needs_input_grad[4 + nsrc + 2 * li] for li in range(n)]
        dsrc = [None] * nsrc

        def layer0(y0, d_a0, bn0, bwdc0, grads, wextra, part, nt_l):
            st = _st(dev)
            off = 0
            for i, (src, c) in enumerate(zip(srcs, chans)):
                if need_src[i]:
                    dx = _empty((b, c, npts), torch.float32, dev)
                    _native.check(_native.timed(
                        _dgrad_kname(lib, b, c, cout0, npts), 2.0 * b * npts * c * cout0, 4.0 * b * npts * (c + 2 * cout0),
                        lambda: lib.istnet_pw_dgrad(b, cin_total, off, c, cout0, npts, 0, w2.data_ptr(), y0.data_ptr(),
                                                    d_a0.data_ptr(), None, 0, None, bn0.data_ptr(), bwdc0.data_ptr(),
                                                    dx.data_ptr(), None, None, None, None, st)), "pw_dgrad(head source)")
                    dsrc[i] = dx
                off += c
            s_cb = None
            if with_mean:
                # per-cloud sums of dY0 from the statistics partials ([cout0][b * tiles], a cloud's tiles contiguous)
                s_cb = part[0].view(cout0, b, nt_l // b).sum(dim=2)          # (cout0, b)
                if need_src[0]:
                    dmean = torch.matmul(w2[:, csum:].t(), s_cb)              # (C, b) = Wb^T . sum_p dY0
                    dsrc[0].add_(dmean.t().unsqueeze(2), alpha=1.0 / npts)
            if need_w[0]:
                dest = _grad_dest(w0, (cout0, cin_total), dev)
                grads[0] = dest.view_as(w0)

                def wjob(wst):
                    red, keep, coff = [], [], 0
                    for src, c in zip(srcs, chans):
                        sp = lib.istnet_pw_wgrad_splits(b, c, cout0, npts)
                        ws = _empty((sp, cout0, c), torch.float32, dev)
                        _native.check(_native.timed(
                            _wgrad_kname(lib, b, c, cout0, npts),
                            2.0 * b * npts * c * cout0, 4.0 * b * npts * (c + 2 * cout0), lambda: lib.istnet_pw_wgrad(
                                b, c, cout0, npts, 0, src.data_ptr(), None, None, y0.data_ptr(), d_a0.data_ptr(), None, 0,
                                None, bn0.data_ptr(), bwdc0.data_ptr(), ws.data_ptr(), wst)), "pw_wgrad(head source)")
                        red.append((cout0 * c, sp, ws.data_ptr(), dest.data_ptr() + 4 * coff, c, cin_total, cout0 * c))
                        keep.append(ws)
                        coff += c
                    _native.reduce_multi(red, wst)     # every source's column block of dW0 in place
                    parts = []
                    if with_mean:
                        parts.append(torch.matmul(s_cb, mean))               # dWb = sum_b (sum_p dY0[b]) (x) mean_b
                        dest[:, csum:].copy_(parts[0])
                    return keep, parts, y0, d_a0, s_cb
                wextra.append(wjob)
            return None
        layer0.takes_partials = True

        with torch.cuda.device(dev):
            dout = dout.contiguous()
            if ctx.pool_mean:          # adjoint of the mean over the points: a dense gradient g[b, c] / N for the stack
                c_last = flat[-3].shape[0]
                dense = _empty((b, c_last, npts), torch.float32, dev)
                _native.check(lib.istnet_expand_rows(b * c_last, npts, dout.data_ptr(), dense.data_ptr(), _st(dev)),
                              "expand_rows")
                dout = dense
            with _head_stack():
                grads, _, _ = _backward_stack(lib, dev, _st(dev), b, cin_total, npts, 1, None, None, False, ys, bns, flat,
                                              None, dout, need_w, True, layer0_hook=layer0)
        out = []
        for li in range(n):
            dw = grads[3 * li]
            out += [dw.view_as(flat[3 * li]) if dw is not None else None, grads[3 * li + 2]]
        return (None, None, None, None, *dsrc, *out)


def pointwise_conv_stack_multi(seq, sources, with_mean=False, pool_mean=False):
    """``seq(cat(sources [+ mean of the single source expanded], dim=1))`` for an ``nn.Sequential`` of
    [Conv1d(k=1) (+ ReLU)]* without building the concatenation on CUDA (FusedMultiSourceBiasMLPFunction); anything the
    fused form does not cover builds the input and runs ``pointwise_conv_stack``."""
    def build():
        x = torch.cat(list(sources), dim=1) if len(sources) > 1 else sources[0]
        if with_mean:
            x = torch.cat([x, x.mean(dim=2, keepdim=True).expand_as(x)], dim=1)
        return x
    mods = list(seq)
    convs, relu_after, i = [], [], 0
    ok = (all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 3 for t in sources)
          and len(sources) <= 6 and all(t.shape[1] % 16 == 0 for t in sources) and sources[0].shape[2] % 32 == 0
          and all(t.shape[0] == sources[0].shape[0] and t.shape[2] == sources[0].shape[2] for t in sources)
          and (not with_mean or len(sources) == 1))
    while ok and i < len(mods):
        m = mods[i]
        if not (isinstance(m, torch.nn.Conv1d) and m.kernel_size == (1,) and m.stride == (1,) and m.padding == (0,)
                and m.groups == 1 and m.bias is not None):
            ok = False
            break
        has_relu = i + 1 < len(mods) and isinstance(mods[i + 1], torch.nn.ReLU)
        convs.append(m)
        relu_after.append(has_relu)
        i += 2 if has_relu else 1
    csum = sum(t.shape[1] for t in sources) * (2 if with_mean else 1)
    if not ok or len(convs) < 2 or not all(relu_after[:-1]) or convs[0].in_channels != csum:
        out = pointwise_conv_stack(seq, build())
        return out.mean(dim=2) if pool_mean else out
    if with_mean:
        # the mean term's backward reads the per-cloud sums of dY0 out of the statistics partials of the layer-1 dgrad,
        # which are laid out [cloud][tile] only on the plain / split-K dgrad paths; the fused small / mid-size backward
        # and the role-split dgrad cut the flattened (cloud, point) axis into chunks that straddle clouds
        lib = _native.lib()
        b, npts = sources[0].shape[0], sources[0].shape[2]
        c0, c1 = convs[0].out_channels, convs[1].out_channels
        if (lib.istnet_pw_bwd_small_ok(c0, c1, npts)
                or lib.istnet_pw_bwd_mid_ok(c0, c1, npts)
                or lib.istnet_pw_dgrad_rs(b, c0, c1, npts, 1)):
            out = pointwise_conv_stack(seq, build())
            return out.mean(dim=2) if pool_mean else out
    params = []
    for m in convs:
        params += [m.weight, m.bias]
    if pool_mean and not relu_after[-1]:
        return FusedMultiSourceBiasMLPFunction.apply(len(sources), with_mean, False, False, *sources, *params).mean(dim=2)
    return FusedMultiSourceBiasMLPFunction.apply(len(sources), with_mean, relu_after[-1], pool_mean, *sources, *params)


def pointwise_conv_stack(seq, x):
    """Run an ``nn.Sequential`` of [Conv1d(k=1) (+ ReLU)]* on x (B, C, N).

    CUDA f32 inputs with N % 32 == 0 take the fused MFMA path; anything else runs ``seq(x)``."""
    mods = list(seq)
    convs, relu_after = [], []
    i = 0
    ok = x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[2] % 32 == 0
    while ok and i < len(mods):
        m = mods[i]
        if not (isinstance(m, torch.nn.Conv1d) and m.kernel_size == (1,) and m.stride == (1,)
                and m.padding == (0,) and m.groups == 1 and m.bias is not None):
            ok = False
            break
        has_relu = i + 1 < len(mods) and isinstance(mods[i + 1], torch.nn.ReLU)
        convs.append(m)
        relu_after.append(has_relu)
        i += 2 if has_relu else 1
    if not ok or not convs or not all(relu_after[:-1]):
        if x.is_cuda:
            _note_fallback(f"per-point conv stack on input {tuple(x.shape)} {x.dtype}: needs float32 (B, C, N) with N % 32 == 0 "
                           "and Conv1d(k=1, bias) [+ ReLU] layers")
        return seq(x)
    params = []
    for m in convs:
        params += [m.weight, m.bias]
    return FusedBiasMLPFunction.apply(x, relu_after[-1], *params)
