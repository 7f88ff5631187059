"""Feature propagation as one fused autograd node (reference pointnet2_modules.py:164-209): ``FusedFPFunction`` -- layer 0 split
by linearity, the interpolation inside the skip product's epilogue, the raw-pair gradient over inverse lists -- ``LazyAct``
(a level's raw output + BatchNorm constants handed to the next level's loaders) and the dispatcher ``fp_level``.
Split out of fused_mlp.py in round 5; the layer stacks, the stream helpers and the process-wide switches stay in
``fused_mlp`` (this module reads them through it at call time), which re-exports the names below."""
import torch
from torch.autograd import Function

from .. import _native
from . import fused_mlp as _fm
from .fused_mlp import (_backward_stack, _bump_counters, _dgrad_kname, _empty, _enter_backward, _forward_stack,
                        _fwd_ld_kname, _grad_dest, _ident_consts, _join_streams, _kname, _wgrad_kname, _layer_args, _note_fallback, _p,
                        _scale_streams, _st)


class FusedFPFunction(Function):
    """Feature propagation: ``mlp(cat([three_interpolate(known_feats, idx, weight), skip]))`` as one node
    (reference pointnet2_modules.py:185-209), with layer 0 split by linearity.  With W0 = [Wa | Wb]:

        y0 = Wa . interp(K) + Wb . S = interp(Wa . K) + Wb . S

    so the product with the interpolated part runs over the m KNOWN points (m = n/2 in the encoder) and the
    interpolation acts on the layer's output width; the concatenated input never exists.  Backward, with
    G' = interp_grad(dY0):  dK = Wa^T . G',  dWa = G' . K^T (both over m points),  dS = Wb^T . dY0,
    dWb = dY0 . S^T (over n points, skip columns only).

    known_feats (B, C2, m), skip (B, C1, n) or None, idx / weight (B, n, 3), csr = _ext.interp_csr(idx, m) or None."""

    @staticmethod
    def forward(ctx, known_feats, skip, idx, weight, csr, training, layers, known_bn, lazy_out, *params):
        # known_bn: None, or the BatchNorm constant block (4, C2) of the stack that produced ``known_feats`` as its RAW last
        #   output (a LazyAct): the loaders of the products over the known points apply relu(scale y + shift) themselves;
        # lazy_out: return (raw last output, its constant block) instead of the activated tensor (see LazyAct)
        from . import _ext
        lib = _native.lib()
        dev = known_feats.device
        known = known_feats.contiguous()
        ksc, ksh = (known_bn[0].data_ptr(), known_bn[1].data_ptr()) if known_bn is not None else (None, None)
        skip_c = skip.contiguous() if skip is not None else None
        b, c2, m = known.shape
        n = idx.shape[1]
        c1 = skip_c.shape[1] if skip_c is not None else 0
        w0, gamma0, beta0 = params[0], params[1], params[2]
        cout0, cin = w0.shape[0], c2 + c1
        w2 = w0.reshape(cout0, cin)
        lay0 = layers[0]
        with torch.cuda.device(dev):
            st = _st(dev)
            zk = _empty((b, cout0, m), torch.float32, dev)
            _native.check(_native.timed(
                _fwd_ld_kname(lib, b, c2, cout0, m), 2.0 * b * m * c2 * cout0,
                4.0 * b * m * (c2 + cout0), lambda: lib.istnet_pw_forward_ld(
                    b, c2, cout0, m, known.data_ptr(), w2.data_ptr(), cin, ksc, ksh, zk.data_ptr(), None, None,
                    st)), "pw_forward_ld(fp)")
            fuse_interp = (skip_c is not None and lib.istnet_pw_forward_cfg(b, c1, cout0, n) == 1
                           and idx.dtype == torch.int32 and idx.is_contiguous() and weight.is_contiguous())
            # no skip features (the finest level): the interpolated product IS y0; its statistics come from the same launch
            interp_stats = (skip_c is None and training and idx.dtype == torch.int32 and idx.is_contiguous()
                            and weight.is_contiguous() and weight.dtype == torch.float32)
            t = None if (fuse_interp or interp_stats) else _ext.three_interpolate(zk, idx, weight)   # (B, cout0, n)
            bn0 = _empty((4, cout0), torch.float32, dev)
            part = None
            if fuse_interp:
                # small launch: the interpolation of zk is evaluated in the epilogue of the skip-connection product
                y0 = _empty((b, cout0, n), torch.float32, dev)
                if training:
                    nt = lib.istnet_pw_forward_ld_tiles(b, c1, cout0, n)
                    part = _empty((2, cout0, nt), torch.float32, dev)
                _native.check(_native.timed(
                    _fwd_ld_kname(lib, b, c1, cout0, n), 2.0 * b * n * c1 * cout0, 4.0 * b * n * (c1 + cout0),
                    lambda: lib.istnet_pw_forward_acc_interp(
                        b, c1, cout0, n, skip_c.data_ptr(), w2.data_ptr() + 4 * c2, cin, zk.data_ptr(), m, idx.data_ptr(),
                        weight.data_ptr(), y0.data_ptr(), _p(part[0]) if training else None,
                        _p(part[1]) if training else None, st)), "pw_forward_acc_interp")
            elif skip_c is not None:
                y0 = _empty((b, cout0, n), torch.float32, dev)
                if training:
                    nt = lib.istnet_pw_forward_ld_tiles(b, c1, cout0, n)
                    part = _empty((2, cout0, nt), torch.float32, dev)
                _native.check(_native.timed(
                    _fwd_ld_kname(lib, b, c1, cout0, n), 2.0 * b * n * c1 * cout0,
                    4.0 * b * n * (c1 + 2 * cout0), lambda: lib.istnet_pw_forward_acc(
                        b, c1, cout0, n, skip_c.data_ptr(), w2.data_ptr() + 4 * c2, cin, t.data_ptr(), y0.data_ptr(),
                        _p(part[0]) if training else None, _p(part[1]) if training else None, st)), "pw_forward_acc")
            elif interp_stats:
                y0 = _empty((b, cout0, n), torch.float32, dev)
                nt = lib.istnet_pw_interp_stats_tiles(b, n)
                part = _empty((2, cout0, nt), torch.float32, dev)
                _native.check(lib.istnet_pw_interp_stats(b, cout0, m, n, zk.data_ptr(), idx.data_ptr(), weight.data_ptr(),
                                                         y0.data_ptr(), part[0].data_ptr(), part[1].data_ptr(), st),
                              "pw_interp_stats")
            else:
                y0 = t
                if training:
                    nt = lib.istnet_pw_bwd_stat_tiles(b, n)
                    part = _empty((2, cout0, nt), torch.float32, dev)
                    _native.check(lib.istnet_pw_channel_stats(b, cout0, n, y0.data_ptr(), part[0].data_ptr(),
                                                              part[1].data_ptr(), st), "pw_channel_stats")
            if training:
                _native.check(lib.istnet_bn_finalize_fwd_nbt(
                    cout0, nt, float(b * n), part[0].data_ptr(), part[1].data_ptr(), gamma0.data_ptr(), beta0.data_ptr(),
                    float(lay0.eps), lay0.momentum_ptr, _p(lay0.running_mean), _p(lay0.running_var),
                    bn0.data_ptr(), lay0.nbt_ptr, st), "bn_finalize_fwd")
            else:
                _native.check(lib.istnet_affine_consts(cout0, gamma0.data_ptr(), beta0.data_ptr(),
                                                       lay0.running_mean.data_ptr(), lay0.running_var.data_ptr(),
                                                       float(lay0.eps), bn0.data_ptr(), st), "affine_consts")
            out, _, ys, bns = _forward_stack(lib, dev, st, b, cin, n, 1, None, None, training, layers, params,
                                             start=(y0, bn0), tail=not lazy_out)
        ctx.training, ctx.dims, ctx.n_layers, ctx.has_skip, ctx.csr = training, (b, c2, c1, m, n), len(layers), \
            skip_c is not None, csr
        ctx.has_known_bn = known_bn is not None
        ctx.save_for_backward(known, skip_c if skip_c is not None else torch.empty(0, device=dev), idx, weight,
                              known_bn if known_bn is not None else torch.empty(0, device=dev), *ys, *bns, *params)
        if lazy_out:
            ctx.mark_non_differentiable(bns[-1])
            ctx.set_materialize_grads(False)      # no zero-filled "gradient" of the constant block (a fill launch per level)
            return ys[-1], bns[-1]
        return out

    @staticmethod
    def backward(ctx, dout, *_unused):
        from . import _ext
        lib = _native.lib()
        b, c2, c1, m, n = ctx.dims
        nl = ctx.n_layers
        sv = ctx.saved_tensors
        if dout is None:                          # (only without materialised gradients: the output was not used)
            dout = torch.zeros_like(sv[5 + nl - 1])
        known, skip, idx, weight = sv[0], (sv[1] if ctx.has_skip else None), sv[2], sv[3]
        known_bn = sv[4] if ctx.has_known_bn else None
        ys, bns, params = sv[5:5 + nl], sv[5 + nl:5 + 2 * nl], sv[5 + 2 * nl:]
        dev = known.device
        _enter_backward(dev)
        cin = c2 + c1
        w0 = params[0]
        cout0 = w0.shape[0]
        w2 = w0.reshape(cout0, cin)
        _native.mark(f"bwd FP(n={n}) start")
        need_known, need_skip = ctx.needs_input_grad[0], ctx.has_skip and ctx.needs_input_grad[1]
        need_w = [ctx.needs_input_grad[9 + 3 * li] for li in range(nl)]
        result = {}

        def layer0(y0, d_a0, bn0, bwdc0, grads, wextra):
            st = _st(dev)
            ident, ibw = _ident_consts(dev, cout0)          # mask always on, dY = g: products of a given dY
            # dY0 = BatchNorm / ReLU backward of dA0.  With the inverse lists of the taps at hand nobody needs it as a
            # tensor: the interpolation gradient forms it per gathered element (istnet_interp_grad_csr_dy) and the GEMM
            # loaders form it from (y0, dA0, constants) as they do in every other layer -- one launch less on the chain.
            raw_pair = ctx.csr is not None
            if raw_pair:
                dy_y, dy_d, dy_bn, dy_bw = y0, d_a0, bn0, bwdc0
            else:
                dy0 = _empty((b, cout0, n), torch.float32, dev)
                _native.check(lib.istnet_pw_dy(b, cout0, n, y0.data_ptr(), d_a0.data_ptr(), bn0.data_ptr(),
                                               bwdc0.data_ptr(), dy0.data_ptr(), st), "pw_dy")
                dy_y, dy_d, dy_bn, dy_bw = dy0, dy0, ident, ibw
            # The skip gradient feeds the set-abstraction backward much later; the interpolation gradient feeds the next
            # (coarser) propagation level at once.  So the skip dgrad leaves the chain: it runs on a side stream beside
            # the interpolation scatter and the known-feature dgrad, joined before this node returns.
            streams = _scale_streams(dev, 2, site="fpskip") if (need_skip and (need_known or need_w[0])) \
                else [torch.cuda.current_stream(dev)] * 2
            if need_skip:
                ds = _empty((b, c1, n), torch.float32, dev)
                with torch.cuda.stream(streams[1]):
                    sst = _st(dev)
                    _native.check(_native.timed(
                        _dgrad_kname(lib, b, c1, cout0, n, dense=True), 2.0 * b * n * c1 * cout0,
                        4.0 * b * n * (c1 + cout0), lambda: lib.istnet_pw_dgrad(
                            b, cin, c2, c1, cout0, n, 0, w2.data_ptr(), dy_y.data_ptr(), dy_d.data_ptr(), None, 0, None,
                            dy_bn.data_ptr(), dy_bw.data_ptr(), ds.data_ptr(), None, None, None, None, sst)),
                        "pw_dgrad(fp skip)")
                result["dskip"] = ds
            gk = None
            if need_known or need_w[0]:
                if raw_pair:
                    gk = _empty((b, cout0, m), torch.float32, dev)                                  # (B, cout0, m)
                    _native.check(lib.istnet_interp_grad_csr_dy(
                        b, cout0, n, m, y0.data_ptr(), d_a0.data_ptr(), bn0.data_ptr(), bwdc0.data_ptr(), weight.data_ptr(),
                        ctx.csr[0].data_ptr(), ctx.csr[1].data_ptr(), gk.data_ptr(), st), "interp_grad_csr_dy")
                else:
                    gk = (_ext.three_interpolate_grad(dy0, idx, weight, m, ctx.csr) if ctx.csr is not None
                          else _ext.three_interpolate_grad(dy0, idx, weight, m))
            if need_known:
                dk = _empty((b, c2, m), torch.float32, dev)
                # (round 5, measured neutral and removed: with the known features a LazyAct, this product's epilogue can leave the
                # BatchNorm-backward sums of dk for the coarser level -- bn_bwd_dense_finalize becomes bn_finalize_bwd there, 3
                # launches of ~13 us -> ~6 us -- but the product itself gets 2-3 us slower: profiles/r05_tail_and_tiles_ab.txt (d))
                _native.check(_native.timed(
                    _dgrad_kname(lib, b, c2, cout0, m, dense=True), 2.0 * b * m * c2 * cout0,
                    4.0 * b * m * (c2 + cout0), lambda: lib.istnet_pw_dgrad(
                        b, cin, 0, c2, cout0, m, 0, w2.data_ptr(), gk.data_ptr(), gk.data_ptr(), None, 0, None,
                        ident.data_ptr(), ibw.data_ptr(), dk.data_ptr(), None, None, None, None, st)),
                    "pw_dgrad(fp known)")
                result["dknown"] = dk
            if need_w[0]:
                dest = _grad_dest(w0, (cout0, cin), dev)
                grads[0] = dest.view_as(w0)
                ksc, ksh = (known_bn[0].data_ptr(), known_bn[1].data_ptr()) if known_bn is not None else (None, None)

                def wjob(wst, reduce_items=None):
                    # reduce_items: the stack's list of split-K sums -- this job adds its two column blocks to it and the
                    # stack sums everything in ONE launch (round 5: a level issued two)
                    sp_a = lib.istnet_pw_wgrad_splits(b, c2, cout0, m)
                    ws_a = _empty((sp_a, cout0, c2), torch.float32, dev)
                    _native.check(_native.timed(
                        _wgrad_kname(lib, b, c2, cout0, m),
                        2.0 * b * m * c2 * cout0, 4.0 * b * m * (c2 + cout0), lambda: lib.istnet_pw_wgrad(
                            b, c2, cout0, m, 0, known.data_ptr(), ksc, ksh, gk.data_ptr(), gk.data_ptr(), None, 0,
                            None, ident.data_ptr(), ibw.data_ptr(), ws_a.data_ptr(), wst)), "pw_wgrad(fp known)")
                    red = [(cout0 * c2, sp_a, ws_a.data_ptr(), dest.data_ptr(), c2, cin, cout0 * c2)]
                    keep = [ws_a]
                    if skip is not None:
                        sp_b = lib.istnet_pw_wgrad_splits(b, c1, cout0, n)
                        ws_b = _empty((sp_b, cout0, c1), torch.float32, dev)
                        _native.check(_native.timed(
                            _wgrad_kname(lib, b, c1, cout0, n),
                            2.0 * b * n * c1 * cout0, 4.0 * b * n * (c1 + cout0), lambda: lib.istnet_pw_wgrad(
                                b, c1, cout0, n, 0, skip.data_ptr(), None, None, dy_y.data_ptr(), dy_d.data_ptr(), None,
                                0, None, dy_bn.data_ptr(), dy_bw.data_ptr(), ws_b.data_ptr(), wst)), "pw_wgrad(fp skip)")
                        red.append((cout0 * c1, sp_b, ws_b.data_ptr(), dest.data_ptr() + 4 * c2, c1, cin, cout0 * c1))
                        keep += [ws_b]
                    if reduce_items is not None:
                        reduce_items.extend(red)
                    else:
                        _native.reduce_multi(red, wst)     # both column blocks of dW0 in place: no concatenation
                    return keep, dy_y, dy_d, dy_bn, dy_bw, gk, known_bn
                wjob.joins_reduce = True
                wextra.append(wjob)
            _join_streams(streams)
            return None

        with torch.cuda.device(dev):
            # the fused mid-size backward kernel takes 128 workgroups by default -- half the chip, because in the
            # set-abstraction phases two or three scale chains run side by side; a feature-propagation level is ONE chain
            # (with the deferred weight gradients beside it): 256 workgroups there, 2.562 -> 2.543 ms on the step
            saved = lib.istnet_pw_get_tuning(8)
            if _fm.FP_BWD_MID_WORKGROUPS:
                lib.istnet_pw_set_tuning(8, _fm.FP_BWD_MID_WORKGROUPS)
            try:
                grads, _, _ = _backward_stack(lib, dev, _st(dev), b, cin, n, 1, None, None, ctx.training, ys, bns, params,
                                              None, dout.contiguous(), need_w, True, layer0_hook=layer0)
            finally:
                lib.istnet_pw_set_tuning(8, saved)
            _native.mark(f"bwd FP(n={n}) chain done")
        return (result.get("dknown"), result.get("dskip"), None, None, None, None, None, None, None, *grads)


class LazyAct:
    """The output of a fused stack BEFORE its last BatchNorm + ReLU: ``raw`` (B, C, n) and the constant block ``bn``
    (4, C: scale, shift, mean, invstd).  Consumers that stage their operands through a loader (the products over the known
    points of the next feature-propagation level) apply relu(scale y + shift) there, so the activated tensor is never
    written or read back and its launch leaves the forward chain.  In the autograd graph ``raw`` STANDS FOR the activated
    values: the gradient that reaches it is the gradient with respect to relu(bn(raw)) -- exactly what the producing
    node's backward expects.  ``materialize()`` gives the activated tensor to a consumer that needs one."""
    __slots__ = ("raw", "bn")

    def __init__(self, raw, bn):
        self.raw, self.bn = raw, bn

    def materialize(self):
        return _MaterializeFn.apply(self.raw, self.bn)


class _MaterializeFn(Function):
    @staticmethod
    def forward(ctx, raw, bn):
        lib = _native.lib()
        b, c, n = raw.shape
        out = _empty((b, c, n), torch.float32, raw.device)
        with torch.cuda.device(raw.device):
            _native.check(lib.istnet_bn_relu_pool(b, c, n, 1, raw.data_ptr(), bn.data_ptr(), out.data_ptr(), 0, None, None,
                                                  _st(raw.device)), "bn_relu_pool")
        return out

    @staticmethod
    def backward(ctx, dout):
        return dout, None          # raw stands for the activated values (see LazyAct)


def fp_level(mlp, known_feats, skip, idx, weight, csr=None, lazy_out=False):
    """``mlp(cat([three_interpolate(known_feats, idx, weight), skip], 1).unsqueeze(-1)).squeeze(-1)`` through the
    fused node when shapes allow; None otherwise (the caller then runs the reference composition)."""
    known_bn = None
    if isinstance(known_feats, LazyAct):
        known_feats, known_bn = known_feats.raw, known_feats.bn
    if not (known_feats.is_cuda and known_feats.dtype == torch.float32):
        return None
    n, m = idx.shape[1], known_feats.shape[2]
    if skip is not None and not (skip.is_cuda and skip.dtype == torch.float32 and skip.shape[2] == n):
        return None
    c1 = skip.shape[1] if skip is not None else 0
    if m % 32 or n % 32 or known_feats.shape[1] % 4 or c1 % 4 or not _fm._fusable_shape(mlp, n, 1):     # (through the module: tests replace it)
        _note_fallback(f"feature propagation with n={n}, m={m}, channels {known_feats.shape[1]}+{c1}: needs n, m % 32 == 0, "
                       "channels % 4 == 0 and a plain conv1x1/BatchNorm/ReLU stack")
        return None
    layers, params = _layer_args(mlp)
    out = FusedFPFunction.apply(known_feats, skip, idx, weight, csr, mlp.training, layers, known_bn, bool(lazy_out), *params)
    if mlp.training:
        _bump_counters(list(mlp))
    return LazyAct(*out) if lazy_out else out
