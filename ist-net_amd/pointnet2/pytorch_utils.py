"""Shared-MLP builders with the reference's module tree and state-dict keys.

Mirror of the parts of model/pointnet2/pytorch_utils.py that the hot path uses
(``SharedMLP`` :25-50, ``BatchNorm{1,2,3}d`` :53-77, ``Conv{1,2,3}d`` :80-250, ``FC`` :253-300,
``BNMomentumScheduler`` :303-330).  A ``SharedMLP([c0, c1, ...], bn=True)`` is
``layer{i}`` = [``conv`` (1x1, bias only when bn is off, kaiming-normal) -> ``normlayer.bn``
(weight 1, bias 0) -> ``activation`` (ReLU, in place)], so a reference checkpoint key such as
``SA_modules.0.mlps.0.layer0.normlayer.bn.running_mean`` loads unchanged.
"""
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


def set_bn_momentum_default(bn_momentum):
    def fn(m):
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            m.momentum = bn_momentum
    return fn


class BNMomentumScheduler(object):
    """Walks the model and sets every BatchNorm's momentum to ``bn_lambda(epoch)``.  [ref :303-330]"""

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
