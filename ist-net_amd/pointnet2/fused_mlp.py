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
from .pytorch_utils import bn_momentum_ptr


if os.environ.get("ISTNET_POISON_ALLOC"):  # debugging aid: expose reads of never-written workspace
    def _empty(shape, dtype, device):
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        if dtype == torch.uint8:
            return torch.full(shape, 255, dtype=dtype, device=device)
        return torch.full(shape, float("nan"), dtype=dtype, device=device)
else:
    def _empty(shape, dtype, device):
        return torch.empty(shape, dtype=dtype, device=device)


STATS = {}         # path counters (tests assert the path they mean)
FALLBACKS = {}     # reason -> number of times a CUDA input took the torch composition instead of the fused kernels


def _note_fallback(reason):
    """A CUDA input that the fused kernels do not cover runs the reference composition on torch / MIOpen: different
    kernels, speed and (within fp32 round-off) numerics.  Counted per reason; the first occurrence warns."""
    import warnings
    FALLBACKS[reason] = FALLBACKS.get(reason, 0) + 1
    if FALLBACKS[reason] == 1:
        warnings.warn(f"istnet_amd: fused MFMA path not taken ({reason}); running the torch composition instead",
                      RuntimeWarning, stacklevel=3)


def _kname(base, cfg, gather=None, pooled=False):
    """Kernel symbol as rocprofv3 prints it, e.g. pw_dgrad_kernel<64, 64, 2, 2> or pw_wgrad_kernel<64, 64, 2, 2, false>.
    ``gather``: None for kernels without the template parameter, else whether the operand loader gathers (wgrad only)."""
    if cfg >= 1000000:       # istnet_pw_wgrad_tile_cfg: the role-split kernel takes dense-input layers of this shape
        cfg -= 1000000
        if base == "pw_wgrad_kernel" and not gather:
            return f"pw_wgrad2_kernel<{cfg // 1000}, {cfg % 1000}, {'true' if pooled else 'false'}>"
        cfg = 64064          # gathered input: pw_wgrad_kernel with its 64 x 64 tiles
    mt, nt = cfg // 1000, cfg % 1000
    if base == "pw_wgrad_kernel" and mt == 32:
        return "pw_wgrad_small_kernel<%s>" % ("true" if gather else "false")
    wm, wn = (1, 4) if mt == 32 else ((4, 1) if mt == 256 else (2, 2))
    tail = "" if gather is None else (", true" if gather else ", false")
    return f"{base}<{mt}, {nt}, {wm}, {wn}{tail}>"


# The kernel names below label launches for the per-launch timing of roofline.py only: without it (_native.TIMING is None) they
# return "" before asking the library anything -- two to three ctypes calls per layer that an eager step does not need.
def _fwd_kname(lib, b, cin, cout, p, plain=True):
    """Kernel an istnet_pw_forward launch runs (split-K / role-split / LDS-tiled, as launch_pw_forward picks)."""
    if _native.TIMING is None:
        return ""
    cfg2 = lib.istnet_pw_forward_cfg(b, cin, cout, p) if plain else 0
    if cfg2 == 1:
        return "pw_fwd_sk_kernel<%d>" % lib.istnet_pw_sk_tm(b, cout, p)
    if cfg2:
        return f"pw_fwd2_kernel<{cfg2 // 1000}, {cfg2 // 100 % 10}, {cfg2 // 10 % 10}, {32 if cfg2 % 10 else 16}, 0>"
    return _kname("pw_fwd_kernel", lib.istnet_pw_tile_cfg(b, cout, p))


def _wgrad_kname(lib, b, cin, cout, p, gather=False, pooled=False):
    if _native.TIMING is None:
        return ""
    return _kname("pw_wgrad_kernel", lib.istnet_pw_wgrad_tile_cfg(b, cin, cout, p), bool(gather), pooled=pooled)


def _fwd_ld_kname(lib, b, cin, cout, p):
    """Kernel an istnet_pw_forward_ld / istnet_pw_forward_acc launch runs: the split-K kernel for small launches."""
    if _native.TIMING is None:
        return ""
    if lib.istnet_pw_forward_cfg(b, cin, cout, p) == 1:
        return "pw_fwd_sk_kernel<%d>" % lib.istnet_pw_sk_tm(b, cout, p)
    return _kname("pw_fwd_kernel", lib.istnet_pw_tile_cfg(b, cout, p))


def _dgrad_kname(lib, b, rows, cout, p, dense=False, stats=False):
    """pw_dgrad_kernel<M_T, N_T, WM, WN, FAST> as launch_pw_dgrad picks it: FAST = every tile interior and the
    reduction length a multiple of the k-tile (kKT = 16)."""
    if _native.TIMING is None:
        return ""
    if dense and lib.istnet_pw_dgrad_sk(b, rows, cout, p):
        return "pw_dgrad_sk_kernel<1>"       # small launch, dense gradient source: K split over the waves, no LDS operands
    if stats and lib.istnet_pw_dgrad_rs(b, rows, cout, p, 1 if dense else 0):
        return "pw_bwd_mid_kernel<8, 4, %s, false>" % ("false" if dense else "true")   # dgrad-only mode
    cfg = lib.istnet_pw_dgrad_tile_cfg(b, rows, p)
    mt, nt = cfg // 1000, cfg % 1000
    fast = rows % mt == 0 and p % nt == 0 and cout % 16 == 0
    return _kname("pw_dgrad_kernel", cfg)[:-1] + (", true>" if fast else ", false>")


def _st(dev):
    return torch.cuda.current_stream(dev).cuda_stream


_CONSTS = {}


def _ident_consts(dev, c):
    """Cached constant blocks for a pass-through layer: bn = (scale 0, shift 1, mean 0, invstd 1) makes the ReLU
    mask always active, bwdc = (1, 0, 0) makes dY = g."""
    key = (dev.index, c)
    if key not in _CONSTS:
        bn = torch.zeros((4, c), dtype=torch.float32, device=dev)
        bn[1].fill_(1.0)
        bn[3].fill_(1.0)
        bwdc = torch.zeros((3, c), dtype=torch.float32, device=dev)
        bwdc[0].fill_(1.0)
        _CONSTS[key] = (bn, bwdc)
    return _CONSTS[key]


# Weight gradients are off the critical path of backward (nothing downstream reads them before the optimizer),
# while the dgrad chain is a sequence of dependent launches.  When the engine allows it, the wgrad GEMMs of a
# stack are issued AFTER its dgrad chain on a dedicated stream (one cross-stream edge per stack) and joined once,
# by an autograd-engine callback at the end of the backward pass: they overlap the dgrad chains of the following
# levels.  Conditions: every parameter's .grad is None (AccumulateGrad then only stores the tensor; an existing
# .grad would be read on the main stream before the join), no kernel timing in progress.  Everything the
# deferred launches read is kept alive until the join.  (A per-layer fork onto a side stream was measured
# slower: 32 extra cross-stream edges per step.)
FP_BWD_MID_WORKGROUPS = int(os.environ.get("ISTNET_FP_BWD_MID_WGS", "256"))   # workgroups of the fused mid-size backward kernel in the FP levels (one chain: the whole chip; the SA phases keep the library's 128)
USE_DEFERRED_WGRAD = os.environ.get("ISTNET_DEFERRED_WGRAD", "1") != "0"    # module attribute; the environment variable only sets its import-time
                             # default (A/B runs).  tests/test_pipeline_gpu.py::test_fallback_paths_agree_with_default flips each
                             # switch once; ist_net.point_branch_side_streams sets and restores this one and USE_SCALE_STREAMS


# Stream priorities were tried too (capture stream and scale streams at priority -1, the wgrad stream at 0): the step
# went from 3.47 to 5.4 ms, so every stream stays at the default priority.
# one wgrad stream for all chains: a stream per concurrent dgrad chain measured 0.2 ms/step SLOWER (the extra
# GEMMs contend with the dependent chains they were meant to stay out of the way of)


class _Deferred:
    streams = {}     # (device index, chain stream id) -> wgrad stream: one per concurrent dgrad chain
    mains = {}       # device index -> stream to join into (the stream the backward nodes run on)
    keep = []        # tensors / closures referenced by launches in flight
    armed = False
    task = -1        # autograd graph-task id the pending work belongs to

    @classmethod
    def arm(cls):
        """Register the end-of-backward join once per backward pass.  If a previous pass died before its callback
        ran (an exception in some backward node), its leftovers are joined and dropped first."""
        task = torch._C._current_graph_task_id()
        if cls.armed and task != cls.task:
            cls.flush()
        if not cls.armed:
            torch.autograd.Variable._execution_engine.queue_callback(cls.flush)
            cls.armed, cls.task = True, task

    @classmethod
    def stream(cls, dev, chain):
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        skey = (key, 0)
        if skey not in cls.streams:
            cls.streams[skey] = torch.cuda.Stream(device=dev)
        return key, cls.streams[skey]

    @classmethod
    def flush(cls):
        """End-of-backward callback.  The engine runs it on the AMBIENT stream of the backward() call, after it has
        joined every leaf stream into that stream -- so that is the stream to join the wgrad stream into.  (Joining into
        the stream of the first fused node that ran, as rounds 1-3 did, is the same thing while every node runs on the
        ambient stream; when that node ran on a forked stream the join landed on a stream the engine had already joined,
        and a capture ended with unjoined work: tools/exp/capture_fork_autograd.py, variant cb_first.)"""
        for (key, _), wstream in cls.streams.items():
            if key in cls.mains:
                with torch.cuda.stream(wstream):
                    _native.mark("wgrad stream drained")
                torch.cuda.current_stream(wstream.device).wait_stream(wstream)
        cls.mains.clear()
        cls.keep.clear()
        cls.armed = False


def deferred_streams(dev):
    """The deferred weight-gradient streams of ``dev`` (parallel.OverlappedFlatReducer orders its collectives after
    them: the kernels that write a parameter's gradient slot are enqueued there before autograd stores ``.grad``)."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    return [s for (k, _), s in _Deferred.streams.items() if k == key]


def _enter_backward(dev):
    """Called at the top of every fused node's backward (on the node's own stream): remembers the stream the
    deferred weight gradients are joined into."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _Deferred.mains:
        _Deferred.mains[key] = torch.cuda.current_stream(dev)


# The per-point MLP stacks of the IST / pose heads (FusedBiasMLPFunction, FusedMultiSourceBiasMLPFunction) may keep deferring
# their weight gradients when the encoders' nodes do not (ist_net.point_branch_side_streams): their wgrad GEMMs (3 ms of the
# full-model step) then run beside the RGB backward instead of delaying the gradient it waits for.
USE_DEFERRED_WGRAD_HEADS = os.environ.get("ISTNET_DEFERRED_WGRAD_HEADS", "0") == "1"
_IN_HEAD_STACK = False


class _head_stack:
    def __enter__(self):
        global _IN_HEAD_STACK
        self.prev, _IN_HEAD_STACK = _IN_HEAD_STACK, USE_DEFERRED_WGRAD_HEADS
    def __exit__(self, *exc):
        global _IN_HEAD_STACK
        _IN_HEAD_STACK = self.prev


def _can_defer(params):
    """Deferral is legal only while AccumulateGrad will just STORE the produced tensors.  A parameter with an
    existing .grad, or one that another node of this pass already produced a gradient for (shared module: the
    engine adds the two tensors on the main stream), forces the launches onto the current stream -- after the
    current stream has waited for whatever the wgrad stream still holds of the earlier producer."""
    if not ((USE_DEFERRED_WGRAD or _IN_HEAD_STACK) and (_native.TIMING is None or _native.TIMING_IN_GRAPH)
            and torch.is_grad_enabled() is False
            and all(getattr(p, "grad", None) is None for p in params)):
        return False
    if any(_Claims.taken_by_other_node(p) for p in params):
        for (key, _), wstream in _Deferred.streams.items():
            if key in _Deferred.mains:
                torch.cuda.current_stream().wait_stream(wstream)
        return False
    # This node becomes the producer of these gradients for the pass -- with or without an optimizer's slot behind them
    # (round 5: _grad_dest claimed only parameters that HAVE a flat-gradient slot, so without FlatAdam a second node of the
    # same pass deferred too and the engine summed two tensors the wgrad stream was still writing: loss = f(m(a)) + f(m(b))
    # gave run-to-run different gradients on small shapes, tools/exp/determinism_small.py)
    for p in params:
        _Claims.claim(p)
    return True


def _p(t):
    return None if t is None else t.data_ptr()


class _Layer:
    """Per-layer constants handed to the autograd function (not differentiable)."""
    __slots__ = ("running_mean", "running_var", "momentum_ptr", "eps", "bias_only", "relu", "nbt_ptr")

    def __init__(self, bn=None, relu=True):
        self.bias_only = bn is None      # conv + bias (+ ReLU) instead of conv + BatchNorm + ReLU
        self.relu = relu                 # False only for the last layer of a bias stack
        self.nbt_ptr = None
        if bn is not None:
            # num_batches_tracked is counted by the launch that updates the running statistics (istnet_bn_finalize_fwd_nbt)
            nbt = bn.num_batches_tracked
            self.nbt_ptr = nbt.data_ptr() if (nbt is not None and nbt.is_cuda and nbt.dtype == torch.int64) else None
            self.running_mean = bn.running_mean
            self.running_var = bn.running_var
            # device address of the module's momentum slot (pytorch_utils._MomentumSlots): the finalize kernels read the
            # momentum from memory, so a captured step follows BNMomentumScheduler.step [ref utils/solver.py:91-92]
            self.momentum_ptr = (bn_momentum_ptr(bn, bn.weight.device)
                                 if (bn.weight.is_cuda and bn.momentum is not None) else None)
            self.eps = bn.eps


class _Gather:
    """Layer-0 input of a set-abstraction scale described by its sources instead of a grouped tensor."""
    __slots__ = ("xyz", "new_xyz", "feat", "idx", "n", "npoint", "nsample", "cfeat", "csr", "compact")

    def __init__(self, xyz, new_xyz, feat, idx, csr=None, compact=None):
        self.xyz, self.new_xyz, self.feat, self.idx = xyz, new_xyz, feat, idx
        self.csr = csr           # (offsets, entries): inverse lists of idx over the n source points, or None
        self.compact = compact   # _ext.BallCompact: evaluate the scale on compact columns (padded repeats once), or None
        self.n = xyz.shape[1]
        self.npoint, self.nsample = idx.shape[1], idx.shape[2]
        self.cfeat = 0 if feat is None else feat.shape[1]


def _forward_stack(lib, dev, st, b, c0, g, s, x, gather, training, layers, params, out_spec=None, start=None, tail=True):
    """Runs all layers + the BN/ReLU/max tail.  Returns (out, arg, ys, bns).  ``tail=False``: the caller applies its own
    tail to the last raw output (ys[-1], bns[-1]); out is None.

    ``out_spec`` = (tensor (B, Ctot, G), channel offset): write the pooled result into that channel slice
    (the MSG concat happens in place) instead of a fresh tensor."""
    p = g * s
    if gather is not None and gather.compact is not None:
        return _forward_stack_compact(lib, dev, st, b, g, s, gather, training, layers, params, out_spec)
    ys, bns = [], []
    cur, cur_c, in_bn = x, c0, None
    pending = None                   # the last layer's finalize, when the tail takes it over
    # layers whose normalisation is a fixed affine map (conv bias; eval-mode BatchNorm): their constant blocks depend
    # on parameters only -- ONE launch for the whole stack, before its first GEMM, instead of one per layer in the chain
    fixed, fixed_items = {}, []
    for li, lay in enumerate(layers):
        if (li == 0 and start is not None) or not (lay.bias_only or not training):
            continue
        cout = params[3 * li].shape[0]
        fixed[li] = _empty((4, cout), torch.float32, dev)
        if lay.bias_only:                    # y + bias: scale 1, shift bias, mean 0, invstd 1
            fixed_items.append((cout, None, params[3 * li + 2].data_ptr(), None, None, 0.0, fixed[li].data_ptr()))
        else:                                # eval-mode BatchNorm: fixed affine map from the running statistics
            fixed_items.append((cout, params[3 * li + 1].data_ptr(), params[3 * li + 2].data_ptr(),
                                lay.running_mean.data_ptr(), lay.running_var.data_ptr(), float(lay.eps),
                                fixed[li].data_ptr()))
    if fixed_items:
        _native.affine_consts_multi(fixed_items, st)
    for li, lay in enumerate(layers):
        w, gamma, beta = params[3 * li], params[3 * li + 1], params[3 * li + 2]
        cout = w.shape[0]
        if li == 0 and start is not None:    # layer 0 (raw output + BN block) was produced by the caller
            ys.append(start[0])
            bns.append(start[1])
            cur, cur_c, in_bn = start[0], cout, start[1]
            continue
        w2 = w.reshape(cout, cur_c)
        bn = fixed[li] if li in fixed else _empty((4, cout), torch.float32, dev)
        if lay.bias_only:
            gamma = None                      # params[3*li+1] is a ones vector, beta is the conv bias
        plain = not (li == 0 and gather is not None)      # istnet_pw_forward (dense input tensor)
        if training and not lay.bias_only:
            nt = lib.istnet_pw_forward_tiles(b, cur_c, cout, p) if plain else lib.istnet_pw_stat_tiles(b, cout, p)
            part = _empty((2, cout, nt), torch.float32, dev)
            ps, pq = part[0].data_ptr(), part[1].data_ptr()
        else:
            nt, ps, pq = 0, None, None
        kname = _fwd_kname(lib, b, cur_c, cout, p, plain)
        flops, nbytes = 2.0 * b * p * cur_c * cout, 4.0 * b * p * (cur_c + cout)
        y = _empty((b, cout, p), torch.float32, dev)
        if li == 0 and gather is not None:
            # layer 0 by linearity: Z = W0[:, 3:] . feat over the n source points (nsample*npoint/n times fewer MACs
            # than over the grouped points), then y0 = Z[:, idx] + W0[:, :3] . (xyz[idx] - centre); an xyz-only
            # layer (level 1) is just the second term
            ga = gather
            z = None
            if ga.cfeat > 0:
                z = _empty((b, cout, ga.n), torch.float32, dev)
                _native.check(_native.timed(
                    _fwd_ld_kname(lib, b, ga.cfeat, cout, ga.n), 2.0 * b * ga.n * ga.cfeat * cout,
                    4.0 * b * ga.n * (ga.cfeat + cout), lambda: lib.istnet_pw_forward_ld(
                        b, ga.cfeat, cout, ga.n, ga.feat.data_ptr(), w2.data_ptr() + 12, cur_c, None, None,
                        z.data_ptr(), None, None, st)), "pw_forward_ld")
            if ps is not None:
                nt = lib.istnet_pw_gather_add_tiles(b, p)
                part = _empty((2, cout, nt), torch.float32, dev)
                ps, pq = part[0].data_ptr(), part[1].data_ptr()
            _native.check(lib.istnet_pw_gather_add(b, ga.n, ga.npoint, ga.nsample, cout, ga.xyz.data_ptr(),
                                                   ga.new_xyz.data_ptr(), ga.idx.data_ptr(), _p(z),
                                                   w2.data_ptr(), cur_c, y.data_ptr(), ps, pq, st), "pw_gather_add")
        else:
            sc, sh = (_p(in_bn[0]), _p(in_bn[1])) if in_bn is not None else (None, None)
            cin_l, src = cur_c, cur
            _native.check(_native.timed(kname, flops, nbytes, lambda: lib.istnet_pw_forward(
                b, cin_l, cout, p, src.data_ptr(), w2.data_ptr(), sc, sh, y.data_ptr(), ps, pq, st)), "pw_forward")
        if li not in fixed:                  # training-mode BatchNorm: batch statistics of this layer's output
            if tail and li == len(layers) - 1 and lay.relu:
                # the stack's tail is a per-channel consumer: it finishes this layer's statistics itself (below)
                pending = (nt, ps, pq, part, gamma, beta, lay)
            else:
                _native.check(lib.istnet_bn_finalize_fwd_nbt(
                    cout, nt, float(b * p), ps, pq, gamma.data_ptr(), beta.data_ptr(), float(lay.eps),
                    lay.momentum_ptr, _p(lay.running_mean), _p(lay.running_var), bn.data_ptr(), lay.nbt_ptr, st),
                    "bn_finalize_fwd")
        ys.append(y)
        bns.append(bn)
        cur, cur_c, in_bn = y, cout, bn
    if not tail:
        return None, None, ys, bns
    if out_spec is None:
        out = _empty((b, cur_c, g), torch.float32, dev)
        out_ptr, out_bstride = out.data_ptr(), 0
    else:
        out, coff = out_spec
        out_ptr, out_bstride = out.data_ptr() + coff * g * 4, out.shape[1] * g
    # arg-max slots (uint8) followed by the raw maxima (f32) in one buffer: see _ymax_ptr
    arg = _empty((_arg_bytes(b * cur_c * g) + 4 * b * cur_c * g,), torch.uint8, dev) if s > 1 else None
    if pending is not None:
        nt_l, ps_l, pq_l, _, gamma_l, beta_l, lay_l = pending
        _native.check(lib.istnet_bn_fin_relu_pool_nbt(
            b, cur_c, g, s, nt_l, float(b * p), ps_l, pq_l, gamma_l.data_ptr(), beta_l.data_ptr(), float(lay_l.eps),
            lay_l.momentum_ptr, _p(lay_l.running_mean), _p(lay_l.running_var), in_bn.data_ptr(), cur.data_ptr(), out_ptr,
            out_bstride, _p(arg), _ymax_ptr(arg, b * cur_c * g), lay_l.nbt_ptr, st), "bn_fin_relu_pool")
    elif s == 1 and not layers[-1].relu:
        _native.check(lib.istnet_affine_apply(b, cur_c, g, 0, cur.data_ptr(), in_bn.data_ptr(), out.data_ptr(), st),
                      "affine_apply")
        # backward of a ReLU-free last layer: gradient mask "always active" (scale 0, shift 1)
        bns[-1] = _ident_consts(dev, cur_c)[0]
    else:
        _native.check(lib.istnet_bn_relu_pool(b, cur_c, g, s, cur.data_ptr(), in_bn.data_ptr(), out_ptr,
                                              out_bstride, _p(arg), _ymax_ptr(arg, b * cur_c * g), st), "bn_relu_pool")
    return out, arg, ys, bns


def _forward_stack_compact(lib, dev, st, b, g, s, ga, training, layers, params, out_spec):
    """_forward_stack on compact columns: activations are (1, C, cap) with the first T columns valid, T on the device.
    Statistics are weighted by the column multiplicities, so BatchNorm sees exactly the sums of the padded evaluation
    (count = b * g * s).  Layer 0 is the split form: Z = W0[:, 3:] . feat over the n source points, then
    y0 = Z[:, source] + W0[:, :3] . (xyz[source] - centre) per compact column."""
    cm = ga.compact
    cap, ncols = cm.cap, cm.ncols_ptr
    ys, bns = [], []
    cur, cur_c, in_bn = None, 3 + ga.cfeat, None
    for li, lay in enumerate(layers):
        w, gamma, beta = params[3 * li], params[3 * li + 1], params[3 * li + 2]
        cout = w.shape[0]
        w2 = w.reshape(cout, cur_c)
        y = _empty((1, cout, cap), torch.float32, dev)
        bn = _empty((4, cout), torch.float32, dev)
        if training:
            nt = cap // 256 if li == 0 else lib.istnet_pw_stat_tiles(1, cout, cap)
            part = _empty((2, cout, nt), torch.float32, dev)
            ps, pq = part[0].data_ptr(), part[1].data_ptr()
        else:
            nt, ps, pq = 0, None, None
        if li == 0:
            z = None
            if ga.cfeat > 0:
                z = _empty((b, cout, ga.n), torch.float32, dev)
                _native.check(_native.timed(
                    _fwd_ld_kname(lib, b, ga.cfeat, cout, ga.n), 2.0 * b * ga.n * ga.cfeat * cout,
                    4.0 * b * ga.n * (ga.cfeat + cout), lambda: lib.istnet_pw_forward_ld(
                        b, ga.cfeat, cout, ga.n, ga.feat.data_ptr(), w2.data_ptr() + 12, cur_c, None, None,
                        z.data_ptr(), None, None, st)), "pw_forward_ld")
            _native.check(lib.istnet_pw_gather_add_cols(
                b, ga.n, g, cap, cout, ga.xyz.data_ptr(), ga.new_xyz.data_ptr(), cm.cidx.data_ptr(), cm.meta.data_ptr(),
                cm.colw.data_ptr(), ncols, _p(z), w2.data_ptr(), cur_c, y.data_ptr(), ps, pq, st), "pw_gather_add_cols")
        else:
            _native.check(lib.istnet_pw_forward_cols(
                cur_c, cout, cap, cur.data_ptr(), w2.data_ptr(), in_bn[0].data_ptr(), in_bn[1].data_ptr(), y.data_ptr(),
                ps, pq, ncols, cm.colw.data_ptr(), st), "pw_forward_cols")
        if training:
            _native.check(lib.istnet_bn_finalize_fwd_nbt(
                cout, nt, float(b * g * s), ps, pq, gamma.data_ptr(), beta.data_ptr(), float(lay.eps),
                lay.momentum_ptr, _p(lay.running_mean), _p(lay.running_var), bn.data_ptr(), lay.nbt_ptr, st), "bn_finalize_fwd")
        else:
            _native.check(lib.istnet_affine_consts(cout, gamma.data_ptr(), beta.data_ptr(), lay.running_mean.data_ptr(),
                                                   lay.running_var.data_ptr(), float(lay.eps), bn.data_ptr(), st),
                          "affine_consts")
        ys.append(y)
        bns.append(bn)
        cur, cur_c, in_bn = y, cout, bn
    if out_spec is None:
        out = _empty((b, cur_c, g), torch.float32, dev)
        out_ptr, out_bstride = out.data_ptr(), 0
    else:
        out, coff = out_spec
        out_ptr, out_bstride = out.data_ptr() + coff * g * 4, out.shape[1] * g
    arg = _empty((_arg_bytes(b * cur_c * g) + 4 * b * cur_c * g,), torch.uint8, dev)
    _native.check(lib.istnet_bn_relu_pool_cols(b, cur_c, g, cap, cur.data_ptr(), in_bn.data_ptr(), cm.gstart.data_ptr(),
                                               out_ptr, out_bstride, arg.data_ptr(), _ymax_ptr(arg, b * cur_c * g), st),
                  "bn_relu_pool_cols")
    return out, arg, ys, bns


# The scales of an MSG level are independent chains of ~10 dependent launches each (GEMM, finalize, ...).
# Run scale i >= 1 on its own stream: one chain's launch gaps and tiny kernels are filled by the other's GEMMs.
# Fork/join discipline: the side stream waits on the main stream before it starts and the main stream joins it
# before the level's result is used, so tensors may cross (allocated in one stream's pool, read by the other).
USE_SCALE_STREAMS = os.environ.get("ISTNET_SCALE_STREAMS", "1") != "0"     # forward and backward of a level, and the FP backward's skip branch
_SCALE_STREAMS = {}


# experiment switch (round 6, profiles/r06_stream_sites.txt): comma-separated sites that do NOT fork -- "fwd<npoint>" / "bwd<npoint>"
# for the scale streams of one SA level (fwd512 ... bwd64), "fpskip" for the skip-branch dgrad of the FP backward.  A fork / join
# pair is two cross-queue dependencies (10-16 us each in a trace against ~1.5 us inside a queue).
NO_FORK_SITES = frozenset(v for v in os.environ.get("ISTNET_NO_FORK", "").split(",") if v)


def _scale_streams(dev, n, site=None):
    """[current stream, side stream 1, ...] for the n scales of a level (side streams only if enabled)."""
    main = torch.cuda.current_stream(dev)
    if (not USE_SCALE_STREAMS or (_native.TIMING is not None and not _native.TIMING_IN_GRAPH) or n < 2
            or (site is not None and site in NO_FORK_SITES)):
        return [main] * n
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    pool = _SCALE_STREAMS.setdefault(key, [])
    while len(pool) < n - 1:
        pool.append(torch.cuda.Stream(device=dev))
    for side in pool[:n - 1]:
        side.wait_stream(main)      # fork before anything of this level is enqueued on the main stream
    return [main] + pool[:n - 1]


def _join_streams(streams):
    for side in streams[1:]:
        if side is not streams[0]:
            streams[0].wait_stream(side)


def _arg_bytes(n):
    return (n + 15) // 16 * 16


def _ymax_ptr(arg, n):
    """Device address of the f32 raw maxima stored behind the n uint8 arg-max slots of `arg` (or None)."""
    return None if arg is None else arg.data_ptr() + _arg_bytes(n)


class _Claims:
    """Parameters whose flat-gradient slot has been handed out during the current backward pass (autograd graph
    task), and the autograd node that took it.  A slot may be written by ONE producer node per pass: when a module
    runs twice in one graph (weight sharing, siamese use, loss = f(m(a)) + f(m(b))), ``param.grad`` is still None at
    the second producer -- AccumulateGrad runs after all uses -- and the engine then sums the producers' outputs,
    which must be distinct tensors."""
    task = -2
    owner = {}     # id(param) -> id of the backward node that produces its gradient in this pass

    @classmethod
    def _sync(cls):
        task = torch._C._current_graph_task_id()
        if task != cls.task:
            cls.task, cls.owner = task, {}
        return id(torch._C._current_autograd_node())

    @classmethod
    def taken_by_other_node(cls, param):
        node = cls._sync()
        return cls.owner.get(id(param), node) != node

    @classmethod
    def claim(cls, param):
        """True when the calling node is (or becomes) the producer of this parameter's gradient in the current pass."""
        node = cls._sync()
        return cls.owner.setdefault(id(param), node) == node


def _grad_dest(param, shape, dev):
    """Where the gradient of `param` is written: its slot of an optimizer's flat gradient buffer when one is
    attached (optim.FlatAdam), no gradient is accumulated yet and no other node of this backward pass took the
    slot already; else a fresh tensor (which autograd then adds to the first producer's)."""
    slot = getattr(param, "_istnet_grad_slot", None)
    if slot is not None and param.grad is None and slot.device == dev:
        n = 1
        for d in shape:
            n *= d
        if slot.numel() == n and _Claims.claim(param):
            return slot.view(shape)
    return _empty(shape, torch.float32, dev)


def _wgrad_job(lib, dev, b, cin, cout, p, ns_arg, use_gather, ga, src, in_bn, y, d_dense, d_pooled, pbs, d_arg, bn,
               bwdc, grad_elems, wparam):
    """Closure launching the split-K wgrad GEMM of one layer on a given stream; returns (elements, splits,
    partials, dw) for the batched reduce.  It owns references to every tensor the launch reads."""
    def launch(wst):
        splits = lib.istnet_pw_wgrad_splits(b, cin, cout, p)
        ws = _empty((splits, cout, cin), torch.float32, dev)
        dw = _grad_dest(wparam, (cout, cin), dev)
        kname = _wgrad_kname(lib, b, cin, cout, p, use_gather, d_dense is None)
        flops = 2.0 * b * p * cin * cout
        dd, dp, da = _p(d_dense), _p(d_pooled), _p(d_arg)
        if use_gather:
            _native.check(_native.timed(
                kname, flops, 4.0 * (b * p * (1 + cout) + grad_elems), lambda: lib.istnet_pw_wgrad_gather(
                    b, ga.n, ga.npoint, ga.nsample, ga.cfeat, cout, ns_arg, ga.xyz.data_ptr(),
                    ga.new_xyz.data_ptr(), _p(ga.feat), ga.idx.data_ptr(), y.data_ptr(), dd, dp, pbs, da,
                    bn.data_ptr(), bwdc.data_ptr(), ws.data_ptr(), wst)), "pw_wgrad_gather")
        else:
            sc, sh = (_p(in_bn[0]), _p(in_bn[1])) if in_bn is not None else (None, None)
            _native.check(_native.timed(
                kname, flops, 4.0 * (b * p * (cin + cout) + grad_elems), lambda: lib.istnet_pw_wgrad(
                    b, cin, cout, p, ns_arg, src.data_ptr(), sc, sh, y.data_ptr(), dd, dp, pbs, da,
                    bn.data_ptr(), bwdc.data_ptr(), ws.data_ptr(), wst)), "pw_wgrad")
        return cout * cin, splits, ws, dw
    return launch


# Set-abstraction levels (0-based) whose scales are evaluated on compact columns (csrc/sa_compact.hip): the repeats that
# pad a ball-query row are computed once and enter every sum over points with their multiplicity.  Level 0 (xyz only,
# 16/16/32 channels) is where the padding is heaviest -- 83 % / 67 % of the slots on the benchmark clouds -- and where it
# pays: 3.31 -> 3.22 ms/step.  Level 1 (68 % / 36 % padded; its tables, lists and weighted dgrad / wgrad / scatter are
# implemented and tested) measured level with the padded evaluation on the same box (3.23 ms): what the fewer columns
# save goes into the dense compact gradient of the pooled layer and the per-column list scatter; level 2 (12 % padded)
# loses 0.25 ms.  ISTNET_COMPACT_LEVELS=0,1 (or assigning this attribute) selects more levels -- denser padding: the
# cube clouds keep 10 % / 39 % at level 1 -- and ISTNET_COMPACT_LEVELS= none.
COMPACT_LEVELS = frozenset(int(v) for v in os.environ.get("ISTNET_COMPACT_LEVELS", "0").split(",") if v != "")
# whether a level of COMPACT_LEVELS keeps its compact columns: "auto" = by the fill ratio the model's first passes see
# (modules.PointNet2MSG._compact_probe), "on" / "off" = always / never
COMPACT_POLICY = os.environ.get("ISTNET_COMPACT_POLICY", "auto")
COMPACT_PROBES = 2          # passes read back before the decision (the warm-up calls that precede any graph capture)
COMPACT_MAX_FILL = 0.6      # distinct columns / padded capacity above which the padded evaluation is the faster one


def _compact_ok(lib, ga, layers, params, needs_backward, needs_feature_grad):
    """Can this scale run on compact columns?  Layer 0 must be the split form (source-point GEMM + gather-add); the
    backward of a scale with input features scatters the layer-0 gradient through the inverse lists of its columns into
    the level-wide feature-gradient GEMM (n <= 4096, n % 4 == 0), which exists only when that gradient is wanted."""
    if ga.compact is None or len(layers) < 2:
        return False
    if ga.cfeat > 0 and (ga.cfeat % 4 or ga.n % 4):
        return False
    if ga.cfeat > 0 and needs_backward:
        return needs_feature_grad and USE_CSR_SCATTER and ga.compact.csr is not None and ga.n <= 4096
    return True


# Round 5 pruned the switches whose off-state was a superseded variant (each was a second code path kept alive by a test):
# the fused small / mid backward kernels, the pooled / dense / in-scatter / in-tail finalizes, the interpolation in the
# skip product's epilogue, the split layer 0, the raw-pair FP gradient and the lazy FP hand-over are simply what runs when
# their shape conditions hold; where a condition fails the older form still runs as the FALLBACK it always was.
USE_CSR_SCATTER = True     # False: LDS-atomic scatter (steps are then not bit-reproducible)


def switch_state():
    """The module-level switches as a hashable value (a captured HIP graph bakes the paths they select in)."""
    g = globals()
    return tuple((k, g[k]) for k in sorted(g) if k.startswith("USE_")) + (COMPACT_LEVELS, FP_BWD_MID_WORKGROUPS, COMPACT_POLICY)


def _dwx_only_job(lib, dev, b, cout, p, ns_arg, ga, y, d_dense, d_pooled, pbs, d_arg, bn, bwdc, wparam):
    """Weight gradient of an xyz-only layer 0: per-(cloud, point chunk) partials of sum_p dY0 * xrel from the
    scatter kernel in its dwx-only mode; the batched reduce sums them into dW0 (cout, 3)."""
    def launch(wst):
        chunks = lib.istnet_pw_dwx_chunks(b, cout, p)
        ws = _empty((b * chunks, cout, 3), torch.float32, dev)
        dw = _grad_dest(wparam, (cout, 3), dev)
        _native.check(lib.istnet_pw_scatter_dy(b, cout, ga.n, p, ns_arg, y.data_ptr(), _p(d_dense), _p(d_pooled), pbs,
                                               _p(d_arg), bn.data_ptr(), bwdc.data_ptr(), ga.idx.data_ptr(), None, 0,
                                               ga.xyz.data_ptr(), ga.new_xyz.data_ptr(), ga.nsample, ws.data_ptr(),
                                               wst), "pw_scatter_dy(dwx)")
        return cout * 3, b * chunks, ws, dw
    return launch


def _reduce_only_job(dev, wparam, cout, cin, splits, ws):
    """wgrad job whose partials already exist (written by the fused backward kernel): only the reduce is left."""
    def launch(wst):
        return cout * cin, splits, ws, _grad_dest(wparam, (cout, cin), dev)
    return launch


def _run_wgrad_jobs(wjobs, wextra, stream):
    """The weight-gradient launches of a stack on ``stream``: its jobs, the extra closures that bring their own split-K sums
    (``joins_reduce``: feature-propagation layer 0), ONE reduce launch for all of them, then the remaining closures."""
    done = [job(stream) for job in wjobs]
    items = [(cnt, splits, ws.data_ptr(), dw.data_ptr()) for cnt, splits, ws, dw in done]
    keep = [fn(stream, items) for fn in wextra if getattr(fn, "joins_reduce", False)]
    if items:
        _native.reduce_multi(items, stream)
    keep += [fn(stream) for fn in wextra if not getattr(fn, "joins_reduce", False)]
    return done, keep


def _backward_stack(lib, dev, st, b, c0, g, s, x, gather, training, ys, bns, params, arg, dout, need_w, need_x,
                    pooled_bstride=0, scatter_out=None, layer0_hook=None):
    """Returns (grads for [w, gamma, beta] * L, gradient w.r.t. the layer-0 input or None).

    With `gather`, the layer-0 input gradient is produced for the feature channels only
    (rows 3..3+cfeat of the grouped tensor), shape (B, cfeat, P).
    ``pooled_bstride``: `dout` is a channel slice (view) of a wider (B, Ctot, G) gradient, consumed in place.
    ``scatter_out`` = (tensor (B, Rtot, n), row offset): the scattered dY0 of the scale goes into those rows and
    the caller finishes the feature gradient for all scales with one GEMM."""
    p = g * s
    if gather is not None and gather.compact is not None:
        return _backward_stack_compact(lib, dev, st, b, g, s, gather, training, ys, bns, params, arg, dout, need_w,
                                       need_x, pooled_bstride, scatter_out)
    n = len(ys)
    grads = [None] * (3 * n)
    pooled = s > 1
    d_dense, d_pooled, d_arg = (None, dout, arg) if pooled else (dout, None, None)
    dx, scattered = None, False
    ntb = lib.istnet_pw_bwd_stat_tiles(b, p)
    fused_part, fused_nt = None, 0   # statistics of layer li already reduced by the dgrad of layer li+1
    wjobs = []                       # weight-gradient launches of the stack: (launch(stream) -> (n, splits, ws, dw))
    wlayers = []                     # layer index of each job
    wextra = []                      # further closures for the wgrad stream (layer0_hook)
    for li in range(n - 1, -1, -1):
        w, gamma = params[3 * li], params[3 * li + 1]
        cout = w.shape[0]
        cin = c0 if li == 0 else params[3 * (li - 1)].shape[0]
        w2 = w.reshape(cout, cin)
        y, bn = ys[li], bns[li]
        ns_arg = s if pooled and li == n - 1 else 0
        dd, dp, da = _p(d_dense), _p(d_pooled), _p(d_arg)
        pbs = pooled_bstride if (pooled and li == n - 1) else 0
        grad_elems = b * cout * (p if dd is not None else p // s)
        part, dense_fin = None, False
        if fused_part is not None:
            part, nt_l = fused_part, fused_nt
        elif ns_arg and not (li == 0 and layer0_hook is not None):
            pass       # statistics and finalize in one launch, below
        elif ns_arg:   # gradient through the max-pool: statistics from the (B, C, G) tensors only
            part, nt_l = _empty((2, cout, b), torch.float32, dev), b
            _native.check(lib.istnet_pw_bwd_stats_pooled(b, cout, g, dp, pbs, _ymax_ptr(d_arg, b * cout * g),
                                                         bn.data_ptr(), part[0].data_ptr(), part[1].data_ptr(), st),
                          "pw_bwd_stats_pooled")
        elif (dd is not None and not ns_arg and b * p <= 65536
              and not (li == 0 and layer0_hook is not None)):
            dense_fin = True       # dense statistics and finalize in one launch, below (small launches: the FP levels)
        else:
            part, nt_l = _empty((2, cout, ntb), torch.float32, dev), ntb
            _native.check(lib.istnet_pw_bwd_stats(b, cout, p, ns_arg, y.data_ptr(), dd, dp, pbs, da, bn.data_ptr(),
                                                  part[0].data_ptr(), part[1].data_ptr(), st), "pw_bwd_stats")
        fused_part = None
        dgamma = _grad_dest(gamma, (cout,), dev)
        dbeta = _grad_dest(params[3 * li + 2], (cout,), dev)
        bwdc = _empty((3, cout), torch.float32, dev)
        # layer 0 of a scale whose gradient leaves through the inverse-list scatter: that kernel derives the constants
        # of its channels from the partials itself (one launch less on the chain)
        scatter_fin = (li == 0 and part is not None and gather is not None and need_x and gather.n <= 4096
                       and layer0_hook is None and USE_CSR_SCATTER and gather.csr is not None
                       and d_dense is not None and p % 4 == 0 and 4 * p <= 65536)
        if scatter_fin:
            pass
        elif dense_fin:
            _native.check(lib.istnet_bn_bwd_dense_finalize(
                b, cout, p, float(b * p), 1 if training else 0, y.data_ptr(), dd, gamma.data_ptr(), bn.data_ptr(),
                dgamma.data_ptr(), dbeta.data_ptr(), bwdc.data_ptr(), st), "bn_bwd_dense_finalize")
        elif part is None:
            _native.check(lib.istnet_bn_bwd_pooled_finalize(
                b, cout, g, float(b * p), 1 if training else 0, dp, pbs, _ymax_ptr(d_arg, b * cout * g), gamma.data_ptr(),
                bn.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), bwdc.data_ptr(), st), "bn_bwd_pooled_finalize")
        else:
            _native.check(lib.istnet_bn_finalize_bwd(
                cout, nt_l, float(b * p), 1 if training else 0, part[0].data_ptr(), part[1].data_ptr(),  # training=False for bias stacks
                gamma.data_ptr(), bn.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), bwdc.data_ptr(), st),
                "bn_finalize_bwd")
        grads[3 * li + 1] = dgamma
        grads[3 * li + 2] = dbeta
        if li == 0 and layer0_hook is not None:
            # the caller finishes layer 0 itself (feature propagation: interpolated + skip input; multi-source heads)
            if getattr(layer0_hook, "takes_partials", False):
                dx = layer0_hook(y, d_dense, bn, bwdc, grads, wextra, part, nt_l)
            else:
                dx = layer0_hook(y, d_dense, bn, bwdc, grads, wextra)
            break
        use_gather = li == 0 and gather is not None
        # layer 0 of a scale inside a fused level: dW0 comes from the scattered dY0 (see FusedSALevelFunction)
        split_w0 = (use_gather and need_w[0] and need_x and scatter_out is not None and gather.n <= 4096
                    and gather.cfeat > 0)
        if li > 0 and need_w[li] and lib.istnet_pw_bwd_small_ok(cin, cout, p):
            # small layer: dA_{l-1}, its statistics partials and the dW partials from ONE pass over (y, g, y_{l-1})
            splits = lib.istnet_pw_bwd_small_splits(b, p)
            ws = _empty((splits, cout, cin), torch.float32, dev)
            dprev = _empty((b, cin, p), torch.float32, dev)
            fused_part, fused_nt = _empty((2, cin, splits), torch.float32, dev), splits
            _native.check(_native.timed(
                "pw_bwd_small_kernel", 4.0 * b * p * cin * cout, 4.0 * (b * p * (2 * cin + cout) + grad_elems),
                lambda: lib.istnet_pw_bwd_small(
                    b, cin, cout, p, ns_arg, w2.data_ptr(), ys[li - 1].data_ptr(), bns[li - 1].data_ptr(),
                    y.data_ptr(), dd, dp, pbs, da, bn.data_ptr(), bwdc.data_ptr(), dprev.data_ptr(),
                    fused_part[0].data_ptr(), fused_part[1].data_ptr(), ws.data_ptr(), st)), "pw_bwd_small")
            wjobs.append(_reduce_only_job(dev, w, cout, cin, splits, ws))
            wlayers.append(li)
            d_dense, d_pooled, d_arg = dprev, None, None
            continue
        if li > 0 and need_w[li] and lib.istnet_pw_bwd_mid_ok(cin, cout, p):
            # mid-size layer (64 / 128 channels): the same one-pass contract, the whole weight matrix in one workgroup
            splits = lib.istnet_pw_bwd_mid_splits(b, cin, cout, p)
            ws = _empty((splits, cout, cin), torch.float32, dev)
            dprev = _empty((b, cin, p), torch.float32, dev)
            fused_part, fused_nt = _empty((2, cin, splits), torch.float32, dev), splits
            _native.check(_native.timed(
                f"pw_bwd_mid_kernel<{cout // 32},{cin // 32},{'false' if dd is not None else 'true'}>", 4.0 * b * p * cin * cout,
                4.0 * (b * p * (2 * cin + cout) + grad_elems),
                lambda: lib.istnet_pw_bwd_mid(
                    b, cin, cout, p, ns_arg, w2.data_ptr(), ys[li - 1].data_ptr(), bns[li - 1].data_ptr(),
                    y.data_ptr(), dd, dp, pbs, da, bn.data_ptr(), bwdc.data_ptr(), dprev.data_ptr(),
                    fused_part[0].data_ptr(), fused_part[1].data_ptr(), ws.data_ptr(), st)), "pw_bwd_mid")
            wjobs.append(_reduce_only_job(dev, w, cout, cin, splits, ws))
            wlayers.append(li)
            d_dense, d_pooled, d_arg = dprev, None, None
            continue
        if use_gather and need_w[0] and gather.cfeat == 0:
            # xyz-only layer 0 (level 1): dW0 = sum_p dY0[:, p] * xrel[p], a reduction over (y0, dA0) -- no GEMM
            wjobs.append(_dwx_only_job(lib, dev, b, cout, p, ns_arg, gather, y, d_dense, d_pooled, pbs, d_arg, bn,
                                       bwdc, w))
            wlayers.append(li)
        elif need_w[li] and not split_w0:
            wjobs.append(_wgrad_job(lib, dev, b, cin, cout, p, ns_arg, use_gather, gather,
                                    x if li == 0 else ys[li - 1], None if li == 0 else bns[li - 1], y,
                                    d_dense, d_pooled, pbs, d_arg, bn, bwdc, grad_elems, w))
            wlayers.append(li)
        if use_gather and need_x and gather.n <= 4096:
            # feature gradient of the scale: scatter dY0 over the ball indices (Cout0 x n per cloud), then
            # the small product W0[:, 3:]^T . G  (see pw_scatter_dy_kernel) -- no (B, C, P) tensor, no big dgrad
            ga = gather
            if scatter_out is None:
                gmat = _empty((b, cout, ga.n), torch.float32, dev)
                gptr, gbs = gmat.data_ptr(), 0
            else:
                gbuf, goff = scatter_out
                gptr, gbs = gbuf.data_ptr() + goff * ga.n * 4, gbuf.shape[1] * ga.n
            dwx = _empty((b, cout, 3), torch.float32, dev) if split_w0 else None
            if USE_CSR_SCATTER and ga.csr is not None and d_dense is not None and p % 4 == 0 and 4 * p <= 65536:
                # atomic-free and deterministic: every source point sums its inverse list from an LDS-staged dY0 row
                if scatter_fin:
                    _native.check(lib.istnet_pw_scatter_dy_csr_fin(
                        b, cout, ga.n, p, y.data_ptr(), dd, bn.data_ptr(), nt_l, float(b * p), 1 if training else 0,
                        part[0].data_ptr(), part[1].data_ptr(), gamma.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                        bwdc.data_ptr(), ga.csr[0].data_ptr(), ga.csr[1].data_ptr(), gptr, gbs, ga.xyz.data_ptr(),
                        ga.new_xyz.data_ptr(), ga.nsample, _p(dwx), st), "pw_scatter_dy_csr_fin")
                else:
                    _native.check(lib.istnet_pw_scatter_dy_csr(
                        b, cout, ga.n, p, y.data_ptr(), dd, bn.data_ptr(), bwdc.data_ptr(), ga.csr[0].data_ptr(),
                        ga.csr[1].data_ptr(), gptr, gbs, ga.xyz.data_ptr(), ga.new_xyz.data_ptr(), ga.nsample, _p(dwx),
                        st), "pw_scatter_dy_csr")
            else:
                _native.check(lib.istnet_pw_scatter_dy(b, cout, ga.n, p, ns_arg, y.data_ptr(), dd, dp, pbs, da,
                                                       bn.data_ptr(), bwdc.data_ptr(), ga.idx.data_ptr(), gptr, gbs,
                                                       ga.xyz.data_ptr(), ga.new_xyz.data_ptr(), ga.nsample, _p(dwx),
                                                       st), "pw_scatter_dy")
            if split_w0:
                grads[0] = dwx          # placeholder: the level node turns it into dW0 (see _finish_layer0_grads)
            if scatter_out is None:
                dx = torch.matmul(w2[:, 3:].t(), gmat)      # (C, Cout0) @ (B, Cout0, n) -> (B, C, n)
            scattered = True
        elif li > 0 or need_x:
            ci_off, rows = (3, gather.cfeat) if use_gather else (0, cin)
            dprev = _empty((b, rows, p), torch.float32, dev)
            if li > 0:   # dprev is dA of layer li-1: reduce its BN-backward statistics in the epilogue
                fused_nt = lib.istnet_pw_dgrad_tiles(b, rows, cout, p, 1 if dd is not None else 0)
                fused_part = _empty((2, rows, fused_nt), torch.float32, dev)
                y_in, bn_in = ys[li - 1].data_ptr(), bns[li - 1].data_ptr()
                pg, pgy = fused_part[0].data_ptr(), fused_part[1].data_ptr()
            else:
                y_in = bn_in = pg = pgy = None
            _native.check(_native.timed(
                _dgrad_kname(lib, b, rows, cout, p, dense=dd is not None, stats=li > 0 and not use_gather), 2.0 * b * p * rows * cout,
                4.0 * (b * p * (rows + cout + (rows if li > 0 else 0)) + grad_elems), lambda: lib.istnet_pw_dgrad(
                    b, cin, ci_off, rows, cout, p, ns_arg, w2.data_ptr(), y.data_ptr(), dd, dp, pbs, da,
                    bn.data_ptr(), bwdc.data_ptr(), dprev.data_ptr(), y_in, bn_in, pg, pgy, st)), "pw_dgrad")
            d_dense, d_pooled, d_arg = dprev, None, None
            if li == 0:
                dx = dprev
    if wjobs or wextra:
        wparams = [params[3 * li] for li in wlayers] + ([params[0]] if wextra else [])
        if _can_defer(wparams):
            # after the chain, on the wgrad stream; joined by the end-of-backward callback
            cur = torch.cuda.current_stream(dev)
            key, wstream = _Deferred.stream(dev, cur)
            _Deferred.mains.setdefault(key, cur)
            wstream.wait_stream(cur)
            with torch.cuda.stream(wstream):
                done, extra_keep = _run_wgrad_jobs(wjobs, wextra, wstream.cuda_stream)
            _Deferred.keep += [wjobs, done, wextra, extra_keep]
            _Deferred.arm()
        else:
            done, _ = _run_wgrad_jobs(wjobs, wextra, st)
        for li, (_, _, _, dw) in zip(wlayers, done):
            grads[3 * li] = dw.view_as(params[3 * li])
    return grads, dx, scattered


def _backward_stack_compact(lib, dev, st, b, g, s, ga, training, ys, bns, params, arg, dout, need_w, need_x,
                            pooled_bstride, scatter_out):
    """_backward_stack on compact columns.  The gradient through the max-pool becomes a dense compact tensor once; the
    layers above layer 0 run the fused small-layer backward (<= 32 channels) or the dgrad / wgrad pair, all with column
    weights; layer 0: the weighted xyz reduction (xyz-only scale), or the weighted scatter over the inverse lists of
    the columns into the level's G buffer (``scatter_out``) -- the level node finishes dW0 and the feature gradient."""
    cm = ga.compact
    cap, ncols, colw = cm.cap, cm.ncols_ptr, cm.colw.data_ptr()
    n = len(ys)
    grads = [None] * (3 * n)
    count = float(b * g * s)
    wjobs, wlayers = [], []
    d_dense, fused_part, fused_nt = None, None, 0
    scattered = False
    for li in range(n - 1, -1, -1):
        w, gamma = params[3 * li], params[3 * li + 1]
        cout = w.shape[0]
        cin = 3 + ga.cfeat if li == 0 else params[3 * (li - 1)].shape[0]
        w2 = w.reshape(cout, cin)
        y, bn = ys[li], bns[li]
        if li == n - 1:      # through the max-pool: the dense compact gradient (statistics + constants below, one launch)
            d_dense = _empty((1, cout, cap), torch.float32, dev)
            _native.check(lib.istnet_pw_pooled_grad_cols(b, cout, g, cap, dout.data_ptr(), pooled_bstride, arg.data_ptr(),
                                                         cm.meta.data_ptr(), ncols, d_dense.data_ptr(), st),
                          "pw_pooled_grad_cols")
        else:
            part, nt_l = fused_part, fused_nt
        dgamma = _grad_dest(gamma, (cout,), dev)
        dbeta = _grad_dest(params[3 * li + 2], (cout,), dev)
        bwdc = _empty((3, cout), torch.float32, dev)
        if li == n - 1:
            # the sums through the max-pool live on the (B, C, G) tensors whatever the column layout (only arg-max columns
            # carry a gradient): the padded path's statistics + finalize launch serves the compact path too (round 5: it ran
            # pw_bwd_stats_pooled and bn_finalize_bwd, two launches on the last chain of the backward pass)
            _native.check(lib.istnet_bn_bwd_pooled_finalize(
                b, cout, g, count, 1 if training else 0, dout.data_ptr(), pooled_bstride, _ymax_ptr(arg, b * cout * g),
                gamma.data_ptr(), bn.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), bwdc.data_ptr(), st),
                "bn_bwd_pooled_finalize")
        else:
            _native.check(lib.istnet_bn_finalize_bwd(
                cout, nt_l, count, 1 if training else 0, part[0].data_ptr(), part[1].data_ptr(), gamma.data_ptr(),
                bn.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), bwdc.data_ptr(), st), "bn_finalize_bwd")
        grads[3 * li + 1], grads[3 * li + 2] = dgamma, dbeta
        if li > 0 and lib.istnet_pw_bwd_small_ok(cin, cout, 256):
            splits = lib.istnet_pw_bwd_small_cols_splits()
            ws = _empty((splits, cout, cin), torch.float32, dev)
            dprev = _empty((1, cin, cap), torch.float32, dev)
            fused_part, fused_nt = _empty((2, cin, splits), torch.float32, dev), splits
            _native.check(lib.istnet_pw_bwd_small_cols(
                cin, cout, cap, w2.data_ptr(), ys[li - 1].data_ptr(), bns[li - 1].data_ptr(),
                y.data_ptr(), d_dense.data_ptr(), bn.data_ptr(), bwdc.data_ptr(), dprev.data_ptr(),
                fused_part[0].data_ptr(), fused_part[1].data_ptr(), ws.data_ptr(), ncols, colw, st), "pw_bwd_small_cols")
            if need_w[li]:
                wjobs.append(_reduce_only_job(dev, w, cout, cin, splits, ws))
                wlayers.append(li)
            d_dense = dprev
        elif li > 0:
            if need_w[li]:
                def wgrad_job(wst, cin=cin, cout=cout, w=w, x=ys[li - 1], bn_in=bns[li - 1], y=y, d=d_dense, bn=bn, bwdc=bwdc):
                    splits = lib.istnet_pw_wgrad_cols_splits(cin, cout)
                    ws = _empty((splits, cout, cin), torch.float32, dev)
                    dw = _grad_dest(w, (cout, cin), dev)
                    _native.check(lib.istnet_pw_wgrad_cols(cin, cout, cap, x.data_ptr(), bn_in[0].data_ptr(),
                                                           bn_in[1].data_ptr(), y.data_ptr(), d.data_ptr(), bn.data_ptr(),
                                                           bwdc.data_ptr(), ws.data_ptr(), ncols, colw, wst), "pw_wgrad_cols")
                    return cout * cin, splits, ws, dw
                wjobs.append(wgrad_job)
                wlayers.append(li)
            dprev = _empty((1, cin, cap), torch.float32, dev)
            fused_nt = lib.istnet_pw_dgrad_stat_tiles(1, cin, cap)
            fused_part = _empty((2, cin, fused_nt), torch.float32, dev)
            _native.check(lib.istnet_pw_dgrad_cols(
                cin, 0, cin, cout, cap, w2.data_ptr(), y.data_ptr(), d_dense.data_ptr(), bn.data_ptr(), bwdc.data_ptr(),
                dprev.data_ptr(), ys[li - 1].data_ptr(), bns[li - 1].data_ptr(), fused_part[0].data_ptr(),
                fused_part[1].data_ptr(), ncols, colw, st), "pw_dgrad_cols")
            d_dense = dprev
        elif ga.cfeat == 0:
            if need_w[0]:
                chunks = lib.istnet_pw_dwx_cols_chunks(cout)

                def dwx_job(wst, cout=cout, chunks=chunks, y0=y, d0=d_dense, bn0=bn, bwdc=bwdc, w=w):
                    ws0 = _empty((chunks, cout, 3), torch.float32, dev)
                    dw = _grad_dest(w, (cout, 3), dev)
                    _native.check(lib.istnet_pw_dwx_cols(cout, cap, y0.data_ptr(), d0.data_ptr(), bn0.data_ptr(),
                                                         bwdc.data_ptr(), cm.cidx.data_ptr(), cm.meta.data_ptr(), colw,
                                                         ncols, ga.xyz.data_ptr(), ga.new_xyz.data_ptr(), ws0.data_ptr(),
                                                         wst), "pw_dwx_cols")
                    return cout * 3, chunks, ws0, dw
                wjobs.append(dwx_job)
                wlayers.append(0)
        elif need_x and scatter_out is not None:
            # G = weighted scatter of dY0 over the inverse lists of the columns, into this scale's rows of the level's
            # buffer; dwx = its xyz-weight partials per cloud (placeholder the level node turns into dW0)
            gbuf, goff = scatter_out
            dwx = _empty((b, cout, 3), torch.float32, dev) if need_w[0] else None
            off, ent = cm.csr
            _native.check(lib.istnet_pw_scatter_dy_csr_cols(
                b, cout, ga.n, g, cap, y.data_ptr(), d_dense.data_ptr(), bn.data_ptr(), bwdc.data_ptr(),
                cm.gstart.data_ptr(), off.data_ptr(), ent.data_ptr(), cm.meta.data_ptr(), colw,
                gbuf.data_ptr() + goff * ga.n * 4, gbuf.shape[1] * ga.n, ga.xyz.data_ptr(), ga.new_xyz.data_ptr(),
                _p(dwx), st), "pw_scatter_dy_csr_cols")
            if need_w[0]:
                grads[0] = dwx
            scattered = True
        else:
            raise RuntimeError("compact set-abstraction scale with input features needs the level-wide feature gradient")
    if wjobs:
        wparams = [params[3 * li] for li in wlayers]
        # An xyz-only scale (level 1) ends the backward pass: nothing follows its chain, so there is nothing for a deferred
        # weight gradient to overlap -- and on the one deferred stream the two scales' xyz reductions and split-K sums ran
        # one after the other behind the last GEMM of the step.  They stay on the scale's own stream (the scales run side by
        # side): 2.565 -> 2.559 ms pipelined, 2.898 -> 2.885 un-pipelined (A/B on one box, profiles/r05_tail_and_tiles_ab.txt).
        chain_ends_here = ga.cfeat == 0
        if not chain_ends_here and _can_defer(wparams):
            cur = torch.cuda.current_stream(dev)
            key, wstream = _Deferred.stream(dev, cur)
            _Deferred.mains.setdefault(key, cur)
            wstream.wait_stream(cur)
            with torch.cuda.stream(wstream):
                done = [job(wstream.cuda_stream) for job in wjobs]
                _native.reduce_multi([(cnt, splits, ws.data_ptr(), dw.data_ptr()) for cnt, splits, ws, dw in done],
                                     wstream.cuda_stream)
            _Deferred.keep += [wjobs, done, ys, bns, cm.tensors()]
            _Deferred.arm()
        else:
            done = [job(st) for job in wjobs]
            _native.reduce_multi([(cnt, splits, ws.data_ptr(), dw.data_ptr()) for cnt, splits, ws, dw in done], st)
        for li, (_, _, _, dw) in zip(wlayers, done):
            grads[3 * li] = dw.view_as(params[3 * li])
    return grads, None, scattered


class FusedSharedMLPFunction(Function):
    """x (B, C0, G, S) -> (B, C_L, G): conv1x1/BN/ReLU stack followed by a max over S."""

    @staticmethod
    def forward(ctx, x, training, layers, *params):
        lib = _native.lib()
        dev = x.device
        b, c0, g, s = x.shape
        x = x.contiguous()
        with torch.cuda.device(dev):
            out, arg, ys, bns = _forward_stack(lib, dev, _st(dev), b, c0, g, s, x, None, training, layers, params)
        ctx.training = training
        ctx.shape = (b, c0, g, s)
        ctx.n_layers = len(layers)
        ctx.save_for_backward(x, arg if arg is not None else torch.empty(0, device=dev), *ys, *bns, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _native.lib()
        b, c0, g, s = ctx.shape
        n = ctx.n_layers
        saved = ctx.saved_tensors
        x, arg = saved[0], saved[1]
        ys, bns, params = saved[2:2 + n], saved[2 + n:2 + 2 * n], saved[2 + 2 * n:]
        dev = x.device
        _enter_backward(dev)
        _native.mark(f"bwd MLP(g={g},s={s}) start")
        need_w = [ctx.needs_input_grad[3 + 3 * li] for li in range(n)]
        with torch.cuda.device(dev):
            grads, dx, _ = _backward_stack(lib, dev, _st(dev), b, c0, g, s, x, None, ctx.training, ys, bns, params,
                                           arg, dout.contiguous(), need_w, ctx.needs_input_grad[0])
        _native.mark(f"bwd MLP(g={g},s={s}) chain done")
        return (dx.view(b, c0, g, s) if dx is not None else None, None, None, *grads)


class FusedSAScaleFunction(Function):
    """One scale of a set-abstraction level without the grouped tensor:
    (xyz (B,n,3), new_xyz (B,npoint,3), features (B,C,n) | None, idx (B,npoint,nsample) i32) -> (B, C_L, npoint).

    Equals ``max_pool(mlp(QueryAndGroup-output))`` (pointnet2_modules.py:61-69, pointnet2_utils.py:348-358):
    the gather, the centroid subtraction and the concat happen in the layer-0 operand loaders; the
    feature gradient is the dgrad of the feature rows followed by the group scatter-add."""

    @staticmethod
    def forward(ctx, features, xyz, new_xyz, idx, training, layers, *params):
        lib = _native.lib()
        dev = xyz.device
        ga = _Gather(xyz.contiguous(), new_xyz.contiguous(), None if features is None else features.contiguous(),
                     idx.contiguous())
        b, g, s = xyz.shape[0], ga.npoint, ga.nsample
        c0 = 3 + ga.cfeat
        with torch.cuda.device(dev):
            out, arg, ys, bns = _forward_stack(lib, dev, _st(dev), b, c0, g, s, None, ga, training, layers, params)
        ctx.training = training
        ctx.shape = (b, c0, g, s)
        ctx.n_layers = len(layers)
        ctx.has_feat = features is not None
        feat_saved = ga.feat if ga.feat is not None else torch.empty(0, device=dev)
        ctx.save_for_backward(feat_saved, ga.xyz, ga.new_xyz, ga.idx, arg, *ys, *bns, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import _ext
        lib = _native.lib()
        b, c0, g, s = ctx.shape
        n = ctx.n_layers
        saved = ctx.saved_tensors
        feat, xyz, new_xyz, idx, arg = saved[:5]
        ys, bns, params = saved[5:5 + n], saved[5 + n:5 + 2 * n], saved[5 + 2 * n:]
        dev = xyz.device
        _enter_backward(dev)
        ga = _Gather(xyz, new_xyz, feat if ctx.has_feat else None, idx)
        need_w = [ctx.needs_input_grad[6 + 3 * li] for li in range(n)]
        need_x = ctx.has_feat and ctx.needs_input_grad[0]
        with torch.cuda.device(dev):
            grads, dxf, scattered = _backward_stack(lib, dev, _st(dev), b, c0, g, s, None, ga, ctx.training, ys,
                                                    bns, params, arg, dout.contiguous(), need_w, need_x)
            dfeat = None
            if need_x:
                dfeat = dxf if scattered else _ext.group_points_grad(dxf.view(b, ga.cfeat, g, s), idx, ga.n)
        return (dfeat, None, None, None, None, None, *grads)


class FusedSALevelFunction(Function):
    """All MSG scales of one set-abstraction level in one autograd node:
    (features (B,C,n) | None, xyz, new_xyz, [idx per scale]) -> (B, sum C_L, npoint).

    Each scale's tail writes its channel slice of the concatenated output directly (no torch.cat); backward
    reads its slice of the incoming gradient in place (no slice copies), scatters dY0 of every scale into one
    (B, sum Cout0, n) buffer and finishes the feature gradient with ONE MFMA GEMM over the concatenated
    layer-0 feature weights -- reference pointnet2_modules.py:60-73 without the glue kernels."""

    @staticmethod
    def forward(ctx, features, xyz, new_xyz, training, scales, csrs, compacts, *tensors):
        # scales: list of per-scale `layers`; csrs: per scale (offsets, entries) inverse lists of idx or None;
        # compacts: per scale _ext.BallCompact (compact-column tables of idx) or None;
        # tensors = [idx_0..idx_{S-1}, params of scale 0, params of scale 1, ...]
        lib = _native.lib()
        dev = xyz.device
        nsc = len(scales)
        idxs = [t.contiguous() for t in tensors[:nsc]]
        xyz, new_xyz = xyz.contiguous(), new_xyz.contiguous()
        feat = None if features is None else features.contiguous()
        b, g = xyz.shape[0], new_xyz.shape[1]
        pos, plist = nsc, []
        for layers in scales:
            plist.append(tensors[pos:pos + 3 * len(layers)])
            pos += 3 * len(layers)
        ctot = sum(pl[-3].shape[0] for pl in plist)
        out = _empty((b, ctot, g), torch.float32, dev)
        saved, meta, coff = [], [], 0
        with torch.cuda.device(dev):
            # the layer-0 weights of the scales stacked for the level-wide feature-gradient product of backward: packed
            # here (one small launch, before the scales fork) so the backward chain does not start with a concatenation
            wcat = None
            if feat is not None and nsc > 1 and ctx.needs_input_grad[0] and xyz.shape[1] <= 4096 and xyz.shape[1] % 4 == 0:
                w0s = [pl[0] for pl in plist]
                if all(w.is_contiguous() for w in w0s):
                    rows, cols = sum(w.shape[0] for w in w0s), 3 + feat.shape[1]
                    if all(a.data_ptr() + 4 * a.numel() == b_.data_ptr() and a.untyped_storage().data_ptr() ==
                           b_.untyped_storage().data_ptr() for a, b_ in zip(w0s[:-1], w0s[1:])):
                        # back to back in one flat parameter buffer (optim.FlatAdam with the module's layout hint): the
                        # stacked matrix is a view, nothing to pack
                        wcat = w0s[0].detach().as_strided((rows, cols), (cols, 1))
                        STATS["wcat_views"] = STATS.get("wcat_views", 0) + 1
                    else:
                        wcat = _empty((rows, cols), torch.float32, dev)
                        _native.pack_words(w0s, wcat, _st(dev))
            streams = _scale_streams(dev, len(scales), site=f"fwd{g}")
            compacts = list(compacts) if compacts is not None else [None] * nsc
            used = []
            for layers, params, idx, stream, cm in zip(scales, plist, idxs, streams, compacts):
                ga = _Gather(xyz, new_xyz, feat, idx, compact=cm)
                if not _compact_ok(lib, ga, layers, params, any(ctx.needs_input_grad), ctx.needs_input_grad[0]):
                    ga.compact = None
                used.append(ga.compact)
                with torch.cuda.stream(stream):
                    _, arg, ys, bns = _forward_stack(lib, dev, _st(dev), b, 3 + ga.cfeat, g, ga.nsample, None, ga,
                                                     training, layers, params, out_spec=(out, coff))
                meta.append((len(layers), ga.nsample, coff, params[-3].shape[0]))
                coff += params[-3].shape[0]
                saved += [arg, *ys, *bns]
            _join_streams(streams)
        ctx.training, ctx.meta, ctx.has_feat = training, meta, feat is not None
        # the feature gradient scatters dY0 over the ball indices: with inverse lists that is atomic-free and
        # deterministic.  Lists not supplied by the caller (PointNet2MSG's geometry pre-pass builds them off the
        # critical path) are built here.
        csrs = list(csrs) if csrs is not None else [None] * nsc
        if USE_CSR_SCATTER and feat is not None and features.requires_grad and xyz.shape[1] <= 4096:
            from . import _ext
            csrs = [c if (c is not None or cm is not None) else _ext.ball_csr(idx, xyz.shape[1])
                    for c, idx, cm in zip(csrs, idxs, used)]     # compact scales scatter through their own lists
        ctx.csrs = csrs
        ctx.compacts = used
        ctx.dims = (b, g, ctot)
        ctx.has_wcat = wcat is not None
        ctx.save_for_backward(feat if feat is not None else torch.empty(0, device=dev), xyz, new_xyz,
                              wcat if wcat is not None else torch.empty(0, device=dev), *idxs, *saved,
                              *tensors[nsc:])
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import _ext
        lib = _native.lib()
        b, g, ctot = ctx.dims
        meta = ctx.meta
        nsc = len(meta)
        sv = ctx.saved_tensors
        feat, xyz, new_xyz = sv[0], sv[1], sv[2]
        wcat_saved = sv[3] if ctx.has_wcat else None
        idxs = sv[4:4 + nsc]
        dev = xyz.device
        _enter_backward(dev)
        _native.mark(f"bwd SA(g={g}) start")
        dout = dout.contiguous()
        pos = 4 + nsc
        per_scale = []
        for (nl, s, coff, clast) in meta:
            arg = sv[pos]
            ys, bns = sv[pos + 1:pos + 1 + nl], sv[pos + 1 + nl:pos + 1 + 2 * nl]
            pos += 1 + 2 * nl
            per_scale.append((arg, ys, bns))
        params_all, ppos = sv[pos:], 0
        cfeat = feat.shape[1] if ctx.has_feat else 0
        n_src = xyz.shape[1]
        need_x = ctx.has_feat and ctx.needs_input_grad[0]
        use_level_gemm = need_x and n_src <= 4096 and n_src % 4 == 0
        cout0_tot = sum(params_all[sum(3 * m[0] for m in meta[:i])].shape[0] for i in range(nsc))
        gbuf = _empty((b, cout0_tot, n_src), torch.float32, dev) if use_level_gemm else None
        grads_all, dfeat, goff, w0f = [], None, 0, []
        base = 7 + nsc   # index of the first parameter among forward()'s arguments
        with torch.cuda.device(dev):
            st = _st(dev)
            streams = _scale_streams(dev, nsc, site=f"bwd{new_xyz.shape[1]}") if (use_level_gemm or not need_x) \
                else [torch.cuda.current_stream(dev)] * nsc
            for (nl, s, coff, clast), (arg, ys, bns), idx, csr, stream, cm in zip(meta, per_scale, idxs, ctx.csrs, streams,
                                                                                      ctx.compacts):
                params = params_all[ppos:ppos + 3 * nl]
                need_w = [ctx.needs_input_grad[base + ppos + 3 * li] for li in range(nl)]
                ppos += 3 * nl
                ga = _Gather(xyz, new_xyz, feat if ctx.has_feat else None, idx, csr=csr, compact=cm)
                cout0 = params[0].shape[0]
                with torch.cuda.stream(stream):
                    grads, dxf, scattered = _backward_stack(
                        lib, dev, _st(dev), b, 3 + cfeat, g, s, None, ga, ctx.training, ys, bns, params, arg,
                        dout[:, coff:coff + clast], need_w, need_x, pooled_bstride=ctot * g,
                        scatter_out=(gbuf, goff) if use_level_gemm else None)
                grads_all += grads
                if use_level_gemm:
                    w0f.append(params[0].reshape(cout0, 3 + cfeat))
                    goff += cout0
                elif need_x:
                    part = dxf if scattered else _ext.group_points_grad(dxf.view(b, cfeat, g, s), idx, n_src)
                    dfeat = part if dfeat is None else dfeat + part
            _join_streams(streams)
            if use_level_gemm:
                # dfeat[b] = [W0f_0^T | W0f_1^T ...] . [G_0; G_1; ...]: the dgrad kernel with identity "BN" constants
                wcat = wcat_saved if wcat_saved is not None else (torch.cat(w0f, dim=0) if nsc > 1 else w0f[0])   # (sum Cout0, 3 + C)
                ident, bwdc = _ident_consts(dev, cout0_tot)                   # mask always on, dY = g
                dfeat = _empty((b, cfeat, n_src), torch.float32, dev)
                _native.check(_native.timed(
                    _dgrad_kname(lib, b, cfeat, cout0_tot, n_src, dense=True),
                    2.0 * b * n_src * cfeat * cout0_tot, 4.0 * b * n_src * (cfeat + cout0_tot), lambda: lib.istnet_pw_dgrad(
                        b, 3 + cfeat, 3, cfeat, cout0_tot, n_src, 0, wcat.data_ptr(), gbuf.data_ptr(), gbuf.data_ptr(),
                        None, 0, None, ident.data_ptr(), bwdc.data_ptr(), dfeat.data_ptr(), None, None, None, None,
                        st)), "pw_dgrad(level)")
                # layer-0 weight gradients of the scales that left a dwx placeholder: dW0[:, 3:] = sum_b G[b].feat[b]^T
                # -- ONE wgrad over the n source points for all scales (gbuf holds every scale's G) -- and
                # dW0[:, :3] = sum_b dwx[b]; off the critical path like the other weight gradients
                w0_slots, pos_g = [], 0
                for i, (nl, _, _, _) in enumerate(meta):
                    gi = sum(3 * m[0] for m in meta[:i])
                    if grads_all[gi] is not None and grads_all[gi].dim() == 3 and grads_all[gi].shape[-1] == 3:
                        w0_slots.append((gi, pos_g, params_all[gi]))
                    pos_g += params_all[gi].shape[0]
                if w0_slots:
                    _finish_layer0_grads(lib, dev, b, cfeat, cout0_tot, n_src, feat, gbuf, ident, bwdc, w0_slots,
                                         grads_all)
            _native.mark(f"bwd SA(g={g}) chains done")
        return (dfeat, None, None, None, None, None, None, *([None] * nsc), *grads_all)


def _finish_layer0_grads(lib, dev, b, cfeat, cout0_tot, n_src, feat, gbuf, ident, bwdc, w0_slots, grads_all):
    """dW0 of every scale of a level from the scattered dY0 (gbuf, (B, sum Cout0, n)) and the per-cloud xyz partials
    left in grads_all as placeholders.  Runs on the deferred-wgrad stream when the engine allows it."""
    params = [p for _, _, p in w0_slots]
    placeholders = [grads_all[gi] for gi, _, _ in w0_slots]
    dests = [_grad_dest(p, (p.shape[0], 3 + cfeat), dev) for p in params]
    for (gi, _, p), d in zip(w0_slots, dests):
        grads_all[gi] = d.view_as(p)

    def work(wst):
        splits = lib.istnet_pw_wgrad_splits(b, cfeat, cout0_tot, n_src)
        ws = _empty((splits, cout0_tot, cfeat), torch.float32, dev)
        _native.check(_native.timed(
            _wgrad_kname(lib, b, cfeat, cout0_tot, n_src),
            2.0 * b * n_src * cfeat * cout0_tot, 4.0 * b * n_src * (cfeat + cout0_tot), lambda: lib.istnet_pw_wgrad(
                b, cfeat, cout0_tot, n_src, 0, feat.data_ptr(), None, None, gbuf.data_ptr(), gbuf.data_ptr(), None, 0,
                None, ident.data_ptr(), bwdc.data_ptr(), ws.data_ptr(), wst)), "pw_wgrad(level G)")
        # one reduce launch writes both column blocks of every scale's dW0 in place: columns 0..2 = sum over the clouds
        # of the xyz partials, columns 3.. = this scale's rows of the level-wide split-K partials
        items, ld = [], 3 + cfeat
        for (gi, row0, p), dwx, dest in zip(w0_slots, placeholders, dests):
            ci = p.shape[0]
            items.append((ci * 3, b, dwx.data_ptr(), dest.data_ptr(), 3, ld, ci * 3))
            items.append((ci * cfeat, splits, ws.data_ptr() + 4 * row0 * cfeat, dest.data_ptr() + 12, cfeat, ld,
                          cout0_tot * cfeat))
        _native.reduce_multi(items, wst)
        return ws

    if _can_defer(params):
        cur = torch.cuda.current_stream(dev)
        key, wstream = _Deferred.stream(dev, cur)
        _Deferred.mains.setdefault(key, cur)
        wstream.wait_stream(cur)
        with torch.cuda.stream(wstream):
            kept = work(wstream.cuda_stream)
        _Deferred.keep += [kept, placeholders, gbuf, feat, dests]
        _Deferred.arm()
    else:
        work(torch.cuda.current_stream(dev).cuda_stream)


def _fusable(mlp, x):
    """True when `mlp` is a plain [conv1x1(no bias) -> BatchNorm2d -> ReLU] x k stack on a CUDA f32 tensor."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        return False
    return _fusable_shape(mlp, x.shape[2], x.shape[3])


def _fusable_shape(mlp, g, s):
    if (g * s) % 32 != 0 or (s > 1 and s not in (4, 8, 16, 32, 64)):
        return False   # the wgrad kernel walks 32-point K chunks that must not straddle clouds
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


class defer_bn_counters:
    """Context manager: the ``num_batches_tracked += 1`` of every fused stack run inside it is issued as ONE
    ``_foreach_add_`` at exit instead of one small launch per stack (PointNet2MSG.forward uses it)."""
    _pending = None

    def __enter__(self):
        self._outer = defer_bn_counters._pending
        if self._outer is None:
            defer_bn_counters._pending = []
        return self

    def __exit__(self, *exc):
        if self._outer is None:
            pending, defer_bn_counters._pending = defer_bn_counters._pending, None
            if pending and exc[0] is None:
                torch._foreach_add_(pending, 1)
        return False


def _bump_counters(units):
    """num_batches_tracked of the stacks that just ran in training mode.  The finalize launches count the batch themselves
    (istnet_bn_finalize_fwd_nbt) whenever the counter is an int64 CUDA tensor -- i.e. always on the fused path; what is left
    here is the odd module whose counter the kernels could not take."""
    counters = [unit.normlayer.bn.num_batches_tracked for unit in units
                if not (unit.normlayer.bn.num_batches_tracked.is_cuda and unit.normlayer.bn.num_batches_tracked.dtype == torch.int64)]
    if not counters:
        return
    if defer_bn_counters._pending is not None:
        defer_bn_counters._pending += counters
    else:
        torch._foreach_add_(counters, 1)


def shared_mlp_maxpool(mlp, x):
    """``max_pool2d(mlp(x), [1, nsample]).squeeze(-1)`` for x (B, C, npoint, nsample) -> (B, C', npoint).

    CUDA tensors of a supported shape run the fused HIP path; anything else (CPU tensors in the
    host-logic tests, exotic module stacks) takes the reference composition with torch ops.
    """
    if not _fusable(mlp, x):
        if x.is_cuda:
            _note_fallback(f"SharedMLP + max-pool on input {tuple(x.shape)} {x.dtype}: needs float32, (npoint * nsample) % 32 == 0, "
                           "nsample in {1, 4, 8, 16, 32, 64} and a plain conv1x1/BatchNorm/ReLU stack")
        act = mlp(x)
        return torch.nn.functional.max_pool2d(act, kernel_size=[1, act.size(3)]).squeeze(-1)
    layers, params = _layer_args(mlp)
    training = mlp.training
    out = FusedSharedMLPFunction.apply(x, training, layers, *params)
    if training:
        _bump_counters(list(mlp))
    return out


def _layer_args(mlp):
    layers, params = [], []
    for unit in mlp:
        bn = unit.normlayer.bn
        layers.append(_Layer(bn))
        params += [unit.conv.weight, bn.weight, bn.bias]
    return layers, params


def sa_scale(grouper, mlp, xyz, new_xyz, features, idx=None):
    """One MSG scale of a set-abstraction level: ``max_pool(mlp(grouper(xyz, new_xyz, features)))``.

    On CUDA, for a plain QueryAndGroup (use_xyz, no normalisation / resampling / extra returns) and a
    fusable SharedMLP, the grouped tensor is never built; otherwise the reference composition runs.
    ``idx`` optionally supplies the ball-query result computed ahead of time (geometry pre-pass)."""
    from . import pointnet2_utils
    plain = (isinstance(grouper, pointnet2_utils.QueryAndGroup) and grouper.use_xyz
             and not (grouper.normalize_xyz or grouper.sample_uniformly or grouper.ret_grouped_xyz
                      or grouper.ret_unique_cnt))
    ok = (plain and xyz.is_cuda and xyz.dtype == torch.float32 and not xyz.requires_grad
          and not new_xyz.requires_grad and grouper.nsample in (4, 8, 16, 32, 64)
          and (features is None or (features.is_cuda and features.dtype == torch.float32)))
    if ok:
        ok = _fusable_shape(mlp, new_xyz.shape[1], grouper.nsample)
    if ok and features is not None and (xyz.shape[1] % 4 or features.shape[1] % 4):
        ok = False          # the source-point GEMM of the split layer 0 takes n % 4 == 0 and channels % 4 == 0
    if not ok:
        return shared_mlp_maxpool(mlp, grouper(xyz, new_xyz, features))
    if idx is None:
        idx = pointnet2_utils.ball_query(grouper.radius, grouper.nsample, xyz, new_xyz)
    layers, params = _layer_args(mlp)
    out = FusedSAScaleFunction.apply(features, xyz, new_xyz, idx, mlp.training, layers, *params)
    if mlp.training:
        _bump_counters(list(mlp))
    return out


def sa_level(groupers, mlps, xyz, new_xyz, features, ball_idx=None, ball_csr=None, ball_compact=None):
    """All scales of a set-abstraction level: ``cat([max_pool(mlp_i(grouper_i(...))) for i], dim=1)``.

    One fused autograd node when every scale qualifies for the gather-fused path (see ``sa_scale``); otherwise the
    scales run one by one and are concatenated with torch.cat as in the reference."""
    from . import pointnet2_utils
    ball_idx = ball_idx if ball_idx is not None else [None] * len(groupers)

    def plain(gr):
        return (type(gr) is pointnet2_utils.QueryAndGroup and gr.use_xyz and gr.nsample in (4, 8, 16, 32, 64)
                and not (gr.normalize_xyz or gr.sample_uniformly or gr.ret_grouped_xyz or gr.ret_unique_cnt))
    ok = (xyz.is_cuda and xyz.dtype == torch.float32 and not xyz.requires_grad and new_xyz is not None
          and not new_xyz.requires_grad
          and (features is None or (features.is_cuda and features.dtype == torch.float32))
          and all(plain(gr) for gr in groupers)
          and all(_fusable_shape(mlp, new_xyz.shape[1], gr.nsample) for gr, mlp in zip(groupers, mlps))
          and (features is None or (xyz.shape[1] % 4 == 0 and features.shape[1] % 4 == 0)))
    if not ok:
        return torch.cat([sa_scale(gr, mlp, xyz, new_xyz, features, idx)
                          for gr, mlp, idx in zip(groupers, mlps, ball_idx)], dim=1)
    idxs = [idx if idx is not None else pointnet2_utils.ball_query(gr.radius, gr.nsample, xyz, new_xyz)
            for gr, idx in zip(groupers, ball_idx)]
    scales, params = [], []
    for mlp in mlps:
        layers, p = _layer_args(mlp)
        scales.append(layers)
        params += p
    training = mlps[0].training
    out = FusedSALevelFunction.apply(features, xyz, new_xyz, training, scales, ball_csr, ball_compact, *idxs, *params)
    if training:
        _bump_counters([unit for mlp in mlps for unit in mlp])
    return out


# ---- the feature-propagation node and the heads' per-point stacks live in their own modules (round 5); the names stay
# importable from here.  (At the bottom: those modules import the stack / stream helpers defined above.)
from .fused_fp import FusedFPFunction, LazyAct, _MaterializeFn, fp_level  # noqa: E402,F401
from .fused_heads import (FusedBiasMLPFunction, FusedMultiSourceBiasMLPFunction, pointwise_conv_stack,  # noqa: E402,F401
                          pointwise_conv_stack_multi)
