"""Shared-MLP builders with the reference's module tree and state-dict keys.

Mirror of the parts of model/pointnet2/pytorch_utils.py that the hot path uses
(``SharedMLP`` :25-50, ``BatchNorm{1,2,3}d`` :53-77, ``Conv{1,2,3}d`` :80-250, ``FC`` :253-300,
``BNMomentumScheduler`` :303-330).  A ``SharedMLP([c0, c1, ...], bn=True)`` is
``layer{i}`` = [``conv`` (1x1, bias only when bn is off, kaiming-normal) -> ``normlayer.bn``
(weight 1, bias 0) -> ``activation`` (ReLU, in place)], so a reference checkpoint key such as
``SA_modules.0.mlps.0.layer0.normlayer.bn.running_mean`` loads unchanged.
"""
import weakref

import torch
import torch.nn as nn


class _NormWrap(nn.Sequential):
    """``<name>bn`` child holding the torch BatchNorm; weight=1, bias=0.  [ref :53-59]"""

    def __init__(self, channels, norm_cls, name=""):
        super().__init__()
        bn = norm_cls(channels)
        nn.init.constant_(bn.weight, 1.0)
        nn.init.constant_(bn.bias, 0.0)
        self.add_module(name + "bn", bn)


class BatchNorm1d(_NormWrap):
    def __init__(self, in_size, name=""):
        super().__init__(in_size, nn.BatchNorm1d, name)


class BatchNorm2d(_NormWrap):
    def __init__(self, in_size, name=""):
        super().__init__(in_size, nn.BatchNorm2d, name)


class BatchNorm3d(_NormWrap):
    def __init__(self, in_size, name=""):
        super().__init__(in_size, nn.BatchNorm3d, name)


class _ConvUnit(nn.Sequential):
    """conv (+ norm) (+ activation); ``preact`` puts norm/activation in front.  [ref :80-134]"""

    _conv_cls = None
    _norm_cls = None
    _unit = None  # default kernel/stride/dilation and padding

    def __init__(self, in_size, out_size, kernel_size=None, stride=None, padding=None, dilation=None,
                 activation=nn.ReLU(inplace=True), bn=False, init=nn.init.kaiming_normal_, bias=True,
                 preact=False, name="", norm_layer=None):
        super().__init__()
        one, zero = self._unit
        conv = self._conv_cls(
            in_size, out_size,
            kernel_size=one if kernel_size is None else kernel_size,
            stride=one if stride is None else stride,
            padding=zero if padding is None else padding,
            dilation=one if dilation is None else dilation,
            bias=bias and not bn)
        init(conv.weight)
        if conv.bias is not None:
            nn.init.constant_(conv.bias, 0.0)

        norm_layer = norm_layer or self._norm_cls
        pre, post = [], []
        if bn:
            (pre if preact else post).append(
                (name + "normlayer", norm_layer(in_size if preact else out_size)))
        if activation is not None:
            (pre if preact else post).append((name + "activation", activation))
        for key, mod in pre + [(name + "conv", conv)] + post:
            self.add_module(key, mod)


class Conv1d(_ConvUnit):
    _conv_cls, _norm_cls, _unit = nn.Conv1d, BatchNorm1d, (1, 0)


class Conv2d(_ConvUnit):
    _conv_cls, _norm_cls, _unit = nn.Conv2d, BatchNorm2d, ((1, 1), (0, 0))


class Conv3d(_ConvUnit):
    _conv_cls, _norm_cls, _unit = nn.Conv3d, BatchNorm3d, ((1, 1, 1), (0, 0, 0))


class FC(nn.Sequential):
    """Linear (+ BatchNorm1d) (+ activation).  [ref :253-300]"""

    def __init__(self, in_size, out_size, activation=nn.ReLU(inplace=True), bn=False, init=None,
                 preact=False, name=""):
        super().__init__()
        fc = nn.Linear(in_size, out_size, bias=not bn)
        if init is not None:
            init(fc.weight)
        if not bn:
            nn.init.constant_(fc.bias, 0.0)
        pre, post = [], []
        if bn:
            (pre if preact else post).append((name + "bn", BatchNorm1d(in_size if preact else out_size)))
        if activation is not None:
            (pre if preact else post).append((name + "activation", activation))
        for key, mod in pre + [(name + "fc", fc)] + post:
            self.add_module(key, mod)


class SharedMLP(nn.Sequential):
    """Stack of 1x1 ``Conv2d`` units over (B, C, npoint, nsample).  [ref :25-50]"""

    def __init__(self, args, bn=False, activation=nn.ReLU(inplace=True), preact=False, first=False,
                 name=""):
        super().__init__()
        for i in range(len(args) - 1):
            plain_head = first and preact and i == 0  # first pre-activated layer: raw conv only
            self.add_module(
                name + "layer{}".format(i),
                Conv2d(args[i], args[i + 1], bn=bn and not plain_head,
                       activation=None if plain_head else activation, preact=preact))


def group_model_params(model, **kwargs):
    """Two optimizer parameter groups: weights (with ``kwargs``) and normalisation / bias terms (weight decay 0).
    [ref :282-299; unused by the reference's own solver, kept for callers that build their optimizer with it]"""
    decayed, plain = [], []
    for name, param in model.named_parameters():
        (plain if ("normlayer" in name or "bias" in name) else decayed).append(param)
    return [dict(params=decayed, **kwargs), dict(params=plain, **{**kwargs, "weight_decay": 0.0})]


class _MomentumSlots:
    """BatchNorm momenta in device memory, one f32 slot per module that runs on the fused kernels.

    The finalize kernels (istnet_bn_finalize_fwd / istnet_bn_finalize_pool_apply) read the momentum through a pointer
    instead of taking it by value, so a step captured in a HIP graph keeps following ``bn.momentum``: the reference
    re-sets every BatchNorm's momentum each iteration (utils/solver.py:91-92 -> pytorch_utils.py:303-330).  The host
    value of a module (``bn.momentum``) stays authoritative; ``ptr`` mirrors it into the module's slot whenever the two
    differ -- with a one-element fill outside a capture, never inside one (a fill recorded in the graph would put the
    old value back on every replay): there the caller must have synced before (``sync_bn_momentum`` /
    ``BNMomentumScheduler.step``, both of which update all slots of a model with one host-to-device copy)."""
    CAP = 4096
    _bufs = {}          # device index -> [device tensor (CAP,), slots in use, host values]
    _free = {}          # device index -> indices of slots whose module was garbage-collected

    @classmethod
    def _buf(cls, dev):
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        if key not in cls._bufs:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("BatchNorm momentum slots cannot be created while a HIP graph is being captured: "
                                   "run one eager step or call pytorch_utils.sync_bn_momentum(model) before the capture")
            cls._bufs[key] = [torch.zeros(cls.CAP, dtype=torch.float32, device=dev), 0, []]
        return key, cls._bufs[key]

    @classmethod
    def _slot(cls, bn, dev):
        """(device key, index) of the module's slot on ``dev``; allocated on first use.  The record carries the owner's id:
        a deep-copied module must not share its source's slot."""
        rec = bn.__dict__.get("_istnet_mslot")
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        if rec is not None and rec[0] == key and rec[2] == id(bn):
            return rec
        key, entry = cls._buf(dev)
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("a BatchNorm module met the fused kernels for the first time inside a HIP-graph capture: "
                               "run one eager step or call pytorch_utils.sync_bn_momentum(model) before the capture")
        free = cls._free.setdefault(key, [])
        if free:
            i = free.pop()
            entry[2][i] = None               # host value of the slot (python float, compared exactly); None forces a sync
        else:
            if entry[1] >= cls.CAP:
                raise RuntimeError(f"more than {cls.CAP} live BatchNorm modules on one device")
            i = entry[1]
            entry[1] += 1
            entry[2].append(None)
        rec = (key, i, id(bn))
        bn.__dict__["_istnet_mslot"] = rec
        weakref.finalize(bn, free.append, i)     # the slot returns to the pool with its module
        return rec

    @classmethod
    def ptr(cls, bn, dev):
        """Device address of the module's momentum slot, holding ``bn.momentum``."""
        key, i, _ = cls._slot(bn, dev)
        buf, _, values = cls._bufs[key]
        want = float(bn.momentum)
        if values[i] != want:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("bn.momentum changed since the last sync and a HIP graph is being captured: call "
                                   "pytorch_utils.sync_bn_momentum(model) (BNMomentumScheduler.step does) before capturing")
            values[i] = want
            buf[i:i + 1].fill_(want)
        return buf.data_ptr() + 4 * i

    @classmethod
    def sync(cls, model):
        """Mirror ``m.momentum`` of every BatchNorm of ``model`` that has (or, for CUDA modules, now gets) a slot; one
        host-to-device copy per device, and only when something changed."""
        dirty = set()
        for m in model.modules():
            if not isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)) or m.momentum is None:
                continue
            dev = m.weight.device if m.weight is not None else (m.running_mean.device if m.running_mean is not None else None)
            if dev is None or dev.type != "cuda":
                continue
            key, i, _ = cls._slot(m, dev)
            values = cls._bufs[key][2]
            want = float(m.momentum)
            if values[i] != want:
                values[i] = want
                dirty.add(key)
        for key in dirty:
            buf, used, values = cls._bufs[key]
            # a FRESH pinned staging tensor per copy: the copy is asynchronous (it queues behind the step in flight), and
            # a staging buffer reused by the next scheduler step would be overwritten before this copy has read it -- the
            # replay in between would then run with the NEXT iteration's momentum (the caching host allocator keeps the
            # block alive until the copy is done).  Slots never synced yet (None) get their value at first use.
            stage = torch.tensor([0.0 if v is None else v for v in values[:used]], dtype=torch.float32).pin_memory()
            buf[:used].copy_(stage, non_blocking=True)


def bn_momentum_ptr(bn, dev):
    return _MomentumSlots.ptr(bn, dev)


def bn_momentum_tensor(bn, dev):
    """The module's device slot as a one-element tensor (for running-statistics updates written with torch ops)."""
    _MomentumSlots.ptr(bn, dev)
    key, i, _ = bn.__dict__["_istnet_mslot"]
    return _MomentumSlots._bufs[key][0][i:i + 1]


def sync_bn_momentum(model):
    """Push every BatchNorm momentum of ``model`` to its device slot (call after changing ``bn.momentum`` by hand when
    the training step is a captured HIP graph; ``BNMomentumScheduler.step`` calls it)."""
    _MomentumSlots.sync(model)


def set_bn_momentum_default(bn_momentum):
    def fn(m):
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            m.momentum = bn_momentum
    return fn


class BNMomentumScheduler(object):
    """Walks the model and sets every BatchNorm's momentum to ``bn_lambda(epoch)``.  [ref :303-330]
    The device-side slots the fused kernels read are refreshed in the same call, so a step replayed from a HIP graph
    sees the new value (the reference steps this scheduler every iteration, utils/solver.py:91-92)."""

    def __init__(self, model, bn_lambda, last_epoch=-1, setter=set_bn_momentum_default):
        if not isinstance(model, nn.Module):
            raise RuntimeError("Class '{}' is not a PyTorch nn Module".format(type(model).__name__))
        self.model = model
        self.setter = setter
        self.lmbd = bn_lambda
        self.step(last_epoch + 1)
        self.last_epoch = last_epoch

    def step(self, epoch=None):
        if epoch is None:
            epoch = self.last_epoch + 1
        self.last_epoch = epoch
        self.model.apply(self.setter(self.lmbd(epoch)))
        sync_bn_momentum(self.model)
