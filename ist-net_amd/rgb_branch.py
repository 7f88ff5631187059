"""RGB branch of IST-Net: ResNet-18 trunk (output stride 8) + pyramid pooling + 3 x (2x upsample, 3x3 conv).

SURVEY.md 8(f) rank 1 -- outside the point-cloud hot path.  The module tree of the reference
(model/modules.py:10-81,225-241 and model/resnet.py:31-60,109-214) with identical child names, so
that a reference checkpoint (`rgb_cam_extractor.model.feats.layer1.0.conv1.weight`, ...) loads
unchanged and the full IST-Net (`ist_net.IST_Net(rgb_extractor=ModifiedResnet())`) can be trained
and benchmarked end to end.  rgb (B,3,H,W) -> (B,128,H,W).  The trunk's convolutions run on
PyTorch-ROCm / MIOpen; on the GPU (channels-last float32) the pyramid module and the three decoder
stages are evaluated in algebraically equivalent forms that do a quarter of the convolution work
(PSPModule, PSPUpsample) over the kernels of include/istnet_rgb.h.

Reference quirks kept on purpose: the `dilation` arguments of layer3 / layer4 are ignored
(resnet.py:153-180 only dilates once the *output_stride* is reached, and that is 32), so both run at
stride 1, dilation 1 on the stride-8 map; `avgpool` / `fc` exist (state-dict keys) but are unused;
the pyramid priors are upsampled with align_corners=False, the decoder with align_corners=True.
No weights are downloaded (the reference fetches resnet18-5c106cde.pth, resnet.py:205-214).
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .pointnet2.pytorch_utils import bn_momentum_ptr, bn_momentum_tensor


# ONE switch for everything this file fuses (round 5; there were eight): the decoder's native PReLU / upsample backward, the
# two-pass BatchNorm + activation kernels of the trunk and the decoder, the pyramid module's linear form, the up-convolution
# split, `final` at the chosen pixels with its MFMA moment passes.  False = the reference's module composition on the
# framework's kernels (model/modules.py:10-81, model/resnet.py:109-202), which is also what CPU tensors and unsupported shapes
# run; tests compare the two.  The trunk's convolutions have their own switch below.
USE_FUSED = os.environ.get("ISTNET_RGB_FUSED", "1") != "0"


class _PReLUFn(torch.autograd.Function):
    """nn.PReLU() with one slope: the framework's forward, include/istnet_rgb.h's backward (one streaming pass, partial
    sums of the slope gradient in a fixed order).  The framework's backward kernel walks the tensor with per-element
    stride arithmetic and ran at 0.35 TB/s on the decoder's channels-last maps -- 24 % of the branch's training step
    (profiles/r02_rgb_branch_breakdown.txt)."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return F.prelu(x, weight)

    @staticmethod
    def backward(ctx, dy):
        from . import _native
        x, weight = ctx.saved_tensors
        fmt = torch.channels_last if (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
                                      and not x.is_contiguous()) else torch.contiguous_format
        x = x.contiguous(memory_format=fmt)
        dy = dy.contiguous(memory_format=fmt)
        dx = torch.empty_like(x, memory_format=fmt)
        lib = _native.lib()
        n = x.numel()
        part = torch.empty(lib.istnet_prelu_bwd_parts(n), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _native.check(lib.istnet_prelu_bwd(n, x.data_ptr(), dy.data_ptr(), weight.data_ptr(), dx.data_ptr(),
                                               part.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream),
                          "prelu_bwd")
        return dx, part.sum().reshape(weight.shape)


class PReLU(nn.PReLU):
    """nn.PReLU (same parameter, same state-dict key) with the native backward on the GPU."""

    def forward(self, x):
        if (USE_FUSED and x.is_cuda and x.dtype == torch.float32 and self.weight.numel() == 1
                and torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad)):
            return _PReLUFn.apply(x, self.weight)
        return super().forward(x)


def _upsample_aligned_forward(x, size):
    """F.interpolate(x, size, mode="bilinear", align_corners=True) of a channels-last float32 CUDA map through
    include/istnet_rgb.h's forward kernel."""
    from . import _native
    b, c, hin, win = x.shape
    y = torch.empty((b, c, size[0], size[1]), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        _native.check(_native.lib().istnet_upsample_bilinear_ac_fwd_nhwc(
            b, c, hin, win, size[0], size[1], x.data_ptr(), y.data_ptr(),
            torch.cuda.current_stream(x.device).cuda_stream), "upsample_bilinear_ac_fwd_nhwc")
    return y


class _UpsampleAlignedFn(torch.autograd.Function):
    """Bilinear upsample with align_corners=True on a channels-last map: the framework's forward, a gather-form backward
    (every input pixel sums its weighted output pixels: no atomics, so the step stays bit-reproducible)."""

    @staticmethod
    def forward(ctx, x, size):
        ctx.in_shape = tuple(x.shape)
        return _upsample_aligned_forward(x, size)

    @staticmethod
    def backward(ctx, dy):
        from . import _native
        b, c, hin, win = ctx.in_shape
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty(ctx.in_shape, dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
        with torch.cuda.device(dy.device):
            _native.check(_native.lib().istnet_upsample_bilinear_ac_bwd_nhwc(
                b, c, hin, win, dy.shape[2], dy.shape[3], dy.data_ptr(), dx.data_ptr(),
                torch.cuda.current_stream(dy.device).cuda_stream), "upsample_bilinear_ac_bwd_nhwc")
        return dx, None


class Upsample2x(nn.Upsample):
    """nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True) with the native backward for channels-last
    float32 maps on the GPU."""

    def __init__(self):
        super().__init__(scale_factor=2, mode="bilinear", align_corners=True)

    def forward(self, x):
        if (USE_FUSED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
                and x.shape[1] % 4 == 0 and x.shape[2] > 1 and x.shape[3] > 1
                and x.is_contiguous(memory_format=torch.channels_last)):
            size = (2 * x.shape[2], 2 * x.shape[3])
            if torch.is_grad_enabled() and x.requires_grad:
                return _UpsampleAlignedFn.apply(x, size)
            return _upsample_aligned_forward(x, size)        # inference: the same forward kernel, no autograd node
        return super().forward(x)


def _conv3x3(cin, cout, stride=1, dilation=1):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=dilation, dilation=dilation, bias=False)


# The trunk's 3x3 / 1x1 convolutions through include/istnet_conv.h.  Measured (profiles/r04_conv_microbench.txt, r04_conv_end_to_end.txt):
# alone, the native forward and backward-data beat MIOpen on the large layers (121 / 120 vs 111 / 115 TFLOP/s at 512 -> 512) and the
# 1x1 ones; inside the training step, beside the two point encoders, the step is the same with either (31.4-31.7 ms); inside
# the inference batch (forward only, the encoder hidden beside the trunk) MIOpen's smaller workgroups leave it more room and
# the batch is 0.3 ms faster with them.  So: on when gradients are recorded; without gradients only if forced ("2").
USE_NATIVE_TRUNK_CONV = os.environ.get("ISTNET_NATIVE_TRUNK_CONV", "1") != "0"
# The native backward-weights product is taken on the 1x1 layers only: alone on the trunk's shapes at B = 32
# (profiles/r04_conv_microbench.txt) the native forward and backward-data win or tie on every stride-1 layer, the native
# backward-weights wins on the 1x1 layers and loses 15 % on the large 3x3 ones, which keep the framework's (MIOpen).  (Rounds 4's
# "all" / "none" modes measured +0.7 ms / +-0 on the step and are gone; istnet_conv_backward_weights itself is tested on
# every trunk shape through the C ABI, tests/test_conv_gpu.py.)


def set_split_precision(enabled):
    """OPT-IN experiment (DESIGN.md "split precision"; off by default, the headline numbers never use it): the trunk's native
    forward and backward-data (stride 1) products on the bf16 matrix pipe -- every fp32 operand split exactly into three bf16
    terms, six bf16 MFMA products per fp32 product, fp32 accumulation.  fp32-class accuracy (the error against float64 is
    0.8-1.0x the exact-fp32 kernel's, tests/test_conv_gpu.py), 1.2-1.6x faster on the trunk's layers.  Process-wide; returns the
    previous setting.  Environment: ISTNET_SPLIT_PRECISION=1."""
    from . import _native
    lib = _native.lib()
    prev = lib.istnet_conv_get_tuning(1)
    _native.check(lib.istnet_conv_set_tuning(1, 1 if enabled else 0), "conv_set_tuning")
    _CONV_GEOM_OK.clear()              # the work-space sizes the guard cached depend on the mode
    global _SPLIT_ON
    _SPLIT_ON = bool(enabled)
    return bool(prev)


def _native_conv_ok(conv, x):
    # without gradients (inference) the exact-fp32 native forward loses 0.3 ms per batch to MIOpen (comment above); the
    # split-precision forward is 1.3x faster than either, so with it on the native path also serves inference
    if not (USE_NATIVE_TRUNK_CONV and (torch.is_grad_enabled() or _SPLIT_ON) and x.is_cuda
            and x.dtype == torch.float32 and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last) and conv.bias is None and conv.groups == 1
            and conv.dilation == (1, 1) and conv.kernel_size[0] == conv.kernel_size[1] and conv.stride[0] == conv.stride[1]
            and conv.padding[0] == conv.padding[1] and isinstance(conv.padding, tuple) and conv.padding_mode == "zeros"
            and conv.weight.dtype == torch.float32 and conv.weight.is_contiguous(memory_format=torch.channels_last)):
        return False
    if x.shape[0] * x.shape[2] * x.shape[3] >= 2 ** 24:
        return False
    key = (x.shape[0], x.shape[2], x.shape[3], conv.in_channels, conv.out_channels, conv.kernel_size[0], conv.kernel_size[1],
           conv.stride[0], conv.padding[0])
    ok = _CONV_GEOM_OK.get(key)
    if ok is None:
        from . import _native
        lib = _native.lib()
        ok = bool(lib.istnet_conv_supported(*key[3:]))
        # a split plan whose work space the library cannot size (>= 2^31 floats, or a geometry it rejects) would make the
        # launch fail inside forward / backward: such a layer keeps the framework's convolution
        ok = ok and lib.istnet_conv_workspace_floats(0, *key) >= 0
        ok = ok and (conv.stride[0] != 1 or lib.istnet_conv_workspace_floats(1, *key) >= 0)
        _CONV_GEOM_OK[key] = ok
    return ok


_CONV_GEOM_OK = {}
_SPLIT_ON = False
if os.environ.get("ISTNET_SPLIT_PRECISION", "0") == "1":
    try:
        set_split_precision(True)
    except RuntimeError:          # library not built yet (import during the build)
        pass


def _conv_workspace(lib, backward_data, args, dev):
    n = lib.istnet_conv_workspace_floats(backward_data, *args)
    return torch.empty((n,), dtype=torch.float32, device=dev) if n > 0 else None


class _ConvFn(torch.autograd.Function):
    """Conv2d (channels-last float32, no bias / groups / dilation) of the ResNet trunk through include/istnet_conv.h: the
    forward and the input gradient as implicit GEMMs on the fp32 matrix cores (reference model/resnet.py:18-25 on cuDNN;
    PyTorch-ROCm on MIOpen), the weight gradient natively on the 1x1 layers, by the framework on the 3x3 ones.  Exact fp32 products,
    fp32 accumulation; deterministic (split-K partial sums are added in a fixed order, no atomics)."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad):
        from . import _native
        lib = _native.lib()
        b, cin, h, w = x.shape
        cout, _, kh, kw = weight.shape
        args = (b, h, w, cin, cout, kh, kw, stride, pad)
        oh, ow = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
        out = torch.empty((b, cout, oh, ow), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        ws = _conv_workspace(lib, 0, args, x.device)
        with torch.cuda.device(x.device):
            _native.check(lib.istnet_conv_forward(*args, x.data_ptr(), weight.data_ptr(), out.data_ptr(),
                                                  ws.data_ptr() if ws is not None else None,
                                                  torch.cuda.current_stream(x.device).cuda_stream), "conv_forward")
        ctx.save_for_backward(x, weight)
        ctx.geom = (stride, pad)
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import _native
        lib = _native.lib()
        x, weight = ctx.saved_tensors
        stride, pad = ctx.geom
        b, cin, h, w = x.shape
        cout, _, kh, kw = weight.shape
        args = (b, h, w, cin, cout, kh, kw, stride, pad)
        dout = dout.contiguous(memory_format=torch.channels_last)
        dx = dw = None
        st = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):
            if ctx.needs_input_grad[0]:
                if stride == 1:
                    dx = torch.empty_like(x, memory_format=torch.channels_last)
                    ws = _conv_workspace(lib, 1, args, x.device)
                    _native.check(lib.istnet_conv_backward_data(*args, dout.data_ptr(), weight.data_ptr(), dx.data_ptr(),
                                                                ws.data_ptr() if ws is not None else None, st), "conv_backward_data")
                else:       # a stride-2 input gradient touches a quarter of the taps per pixel: the gather form wastes the rest
                    dx = torch.ops.aten.convolution_backward(dout, x, weight, None, [stride, stride], [pad, pad], [1, 1], False,
                                                             [0, 0], 1, [True, False, False])[0]
            if ctx.needs_input_grad[1]:
                splits = lib.istnet_conv_wrw_splits(*args)
                native = splits > 0 and kh == 1
                if native:
                    part = torch.empty((splits, weight.numel()), dtype=torch.float32, device=x.device)
                    dw = torch.empty_like(weight, memory_format=torch.channels_last)
                    _native.check(lib.istnet_conv_backward_weights(*args, x.data_ptr(), dout.data_ptr(), part.data_ptr(),
                                                                   dw.data_ptr(), st), "conv_backward_weights")
                else:
                    dw = torch.ops.aten.convolution_backward(dout, x, weight, None, [stride, stride], [pad, pad], [1, 1], False,
                                                             [0, 0], 1, [False, True, False])[1]
        return dx, dw, None, None


def _conv(conv, x):
    """conv(x) for a trunk convolution: include/istnet_conv.h when the layer and the tensor qualify, the module otherwise."""
    if _native_conv_ok(conv, x):
        return _ConvFn.apply(x, conv.weight, conv.stride[0], conv.padding[0])
    return conv(x)


_COUNTER_SCOPE = None     # inside ModifiedResnet.forward: the BatchNorm batch counters of the fused sites, bumped by ONE launch


def _bump_batch_counter(bn):
    """``num_batches_tracked += 1`` of a BatchNorm whose statistics a fused kernel just updated (torch does this in
    BatchNorm.forward).  Twenty-three four-microsecond launches on the RGB branch's stream when done one by one; inside
    the extractor's forward they are collected and added by one multi-tensor launch at its end."""
    if not (bn.track_running_stats and bn.num_batches_tracked is not None):
        return
    if _COUNTER_SCOPE is not None:
        _COUNTER_SCOPE.append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked.add_(1)




class _BnReluFn(torch.autograd.Function):
    """BatchNorm2d (training statistics) [+ residual] -> ReLU of a channels-last map: the two normalisation sites of a
    ResNet basic block and the stem (reference model/resnet.py:48-65,166-168).  The framework's sequence is MIOpen's
    three-kernel BatchNorm, an add and a clamp forward (8 passes over the map with the residual, 5 without) and threshold
    + three-kernel BatchNorm backward (8); here: statistics + apply forward (4 / 3), statistics + apply backward (7 / 5),
    the same kernels as the decoder stages (include/istnet_rgb.h: slope 0 is ReLU; with a residual the sign of the
    activation's argument is read from the saved output and the masked gradient g, which IS the identity branch's
    gradient, is written by the statistics pass).  ``zero`` / ``one``: one-element constants on the device."""

    @staticmethod
    def forward(ctx, y, gamma, beta, res, zero, one, running_mean, running_var, momentum, eps):
        from . import _native
        lib = _native.lib()
        b, c, h, w = y.shape
        dev = y.device
        rows = b * h * w
        nparts = lib.istnet_nhwc_stat_parts(rows)
        part = torch.empty((2, c, nparts), dtype=torch.float32, device=dev)
        bn = torch.empty((4, c), dtype=torch.float32, device=dev)
        z = torch.empty_like(y, memory_format=torch.channels_last)
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            _native.check(lib.istnet_nhwc_channel_stats(rows, c, y.data_ptr(), part[0].data_ptr(), part[1].data_ptr(), st),
                          "nhwc_channel_stats")
            _native.check(lib.istnet_bn_finalize_fwd(
                c, nparts, float(rows), part[0].data_ptr(), part[1].data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps),
                momentum, running_mean.data_ptr() if running_mean is not None else None,
                running_var.data_ptr() if running_var is not None else None, bn.data_ptr(), st), "bn_finalize_fwd")
            if res is None:
                _native.check(lib.istnet_nhwc_bn_prelu_apply(b, h * w, c, y.data_ptr(), bn.data_ptr(), zero.data_ptr(), None,
                                                             z.data_ptr(), st), "nhwc_bn_prelu_apply")
            else:
                _native.check(lib.istnet_nhwc_bn_act_res_apply(b, h * w, c, y.data_ptr(), bn.data_ptr(), zero.data_ptr(),
                                                               res.data_ptr(), z.data_ptr(), st), "nhwc_bn_act_res_apply")
        ctx.has_res = res is not None
        ctx.save_for_backward(y, gamma, bn, zero, one, z if res is not None else torch.empty(0, device=dev))
        return z

    @staticmethod
    def backward(ctx, dz):
        from . import _native
        lib = _native.lib()
        y, gamma, bn, zero, one, z = ctx.saved_tensors
        b, c, h, w = y.shape
        dev = y.device
        rows = b * h * w
        dz = dz.contiguous(memory_format=torch.channels_last)
        nparts = lib.istnet_nhwc_stat_parts(rows)
        part = torch.empty((2, c, nparts), dtype=torch.float32, device=dev)
        pslope = torch.empty((nparts,), dtype=torch.float32, device=dev)
        dgamma = torch.empty((c,), dtype=torch.float32, device=dev)
        dbeta = torch.empty((c,), dtype=torch.float32, device=dev)
        bwdc = torch.empty((3, c), dtype=torch.float32, device=dev)
        dy = torch.empty_like(y, memory_format=torch.channels_last)
        g = torch.empty_like(y, memory_format=torch.channels_last) if ctx.has_res else None
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            if g is None:
                _native.check(lib.istnet_nhwc_bn_prelu_bwd_stats(b, h * w, c, y.data_ptr(), dz.data_ptr(), bn.data_ptr(),
                                                                 zero.data_ptr(), None, part[0].data_ptr(), part[1].data_ptr(),
                                                                 pslope.data_ptr(), st), "nhwc_bn_prelu_bwd_stats")
            else:
                _native.check(lib.istnet_nhwc_bn_act_res_bwd_stats(b, h * w, c, y.data_ptr(), dz.data_ptr(), z.data_ptr(),
                                                                   bn.data_ptr(), zero.data_ptr(), g.data_ptr(),
                                                                   part[0].data_ptr(), part[1].data_ptr(), pslope.data_ptr(),
                                                                   st), "nhwc_bn_act_res_bwd_stats")
            _native.check(lib.istnet_bn_finalize_bwd(c, nparts, float(rows), 1, part[0].data_ptr(), part[1].data_ptr(),
                                                     gamma.data_ptr(), bn.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                                     bwdc.data_ptr(), st), "bn_finalize_bwd")
            if g is None:
                _native.check(lib.istnet_nhwc_bn_prelu_bwd_apply(b, h * w, c, y.data_ptr(), dz.data_ptr(), bn.data_ptr(),
                                                                 bwdc.data_ptr(), zero.data_ptr(), None, dy.data_ptr(), st),
                              "nhwc_bn_prelu_bwd_apply")
            else:                   # g already carries the activation's mask: slope 1 makes the apply pass take it as it is
                _native.check(lib.istnet_nhwc_bn_prelu_bwd_apply(b, h * w, c, y.data_ptr(), g.data_ptr(), bn.data_ptr(),
                                                                 bwdc.data_ptr(), one.data_ptr(), None, dy.data_ptr(), st),
                              "nhwc_bn_prelu_bwd_apply")
        return dy, dgamma, dbeta, g, None, None, None, None, None, None


def _bn_relu(bn, y, res, owner, identity=False):
    """relu(bn(y) [+ res]) -- by _BnReluFn when the map is a channels-last float32 CUDA tensor and ``bn`` normalises with
    batch statistics, by the framework's modules otherwise (eval mode, CPU, other layouts).  ``identity``: bn(y) alone (the
    BatchNorm of a block's downsample branch, reference model/resnet.py:139-143): the same two passes with slope 1."""
    c = y.shape[1] if y.dim() == 4 else 0
    if (USE_FUSED and y.is_cuda and y.dtype == torch.float32 and y.dim() == 4 and bn.training and bn.affine
            and bn.momentum is not None and c % 4 == 0 and 4 <= c <= 1024 and torch.is_grad_enabled()
            and y.is_contiguous(memory_format=torch.channels_last)
            and (res is None or (res.shape == y.shape and res.dtype == torch.float32))):
        if res is not None:
            res = res.contiguous(memory_format=torch.channels_last)
        _bump_batch_counter(bn)
        return _BnReluFn.apply(y, bn.weight, bn.bias, res, owner._one if identity else owner._zero, owner._one,
                               bn.running_mean if bn.track_running_stats else None,
                               bn.running_var if bn.track_running_stats else None, bn_momentum_ptr(bn, y.device), bn.eps)
    out = bn(y)
    if identity:
        return out
    if res is not None:
        out = out + res
    return torch.relu(out)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = _conv3x3(inplanes, planes, stride=stride, dilation=dilation)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv3x3(planes, planes, dilation=dilation)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride
        self.register_buffer("_zero", torch.zeros(1), persistent=False)    # ReLU as a PReLU slope / identity slope
        self.register_buffer("_one", torch.ones(1), persistent=False)

    def forward(self, x):
        out = _bn_relu(self.bn1, _conv(self.conv1, x), None, self)
        return _bn_relu(self.bn2, _conv(self.conv2, out), x if self.downsample is None else self._shortcut(x), self)

    def _shortcut(self, x):
        ds = self.downsample
        if isinstance(ds, nn.Sequential) and len(ds) == 2 and isinstance(ds[1], nn.BatchNorm2d):
            return _bn_relu(ds[1], _conv(ds[0], x), None, self, identity=True)
        return ds(x)


class ResNet(nn.Module):
    """Trunk returning (layer4 output, layer3 output), both at 1/8 resolution.  [ref resnet.py:109-202]"""

    def __init__(self, layers=(2, 2, 2, 2), num_classes=1000, output_stride=32):
        super().__init__()
        self._inplanes, self._stride, self._dilation, self._output_stride = 64, 4, 1, output_stride
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._stage(64, layers[0])
        self.layer2 = self._stage(128, layers[1], stride=2)
        self.layer3 = self._stage(256, layers[2])   # reference passes dilation=2: ignored there
        self.layer4 = self._stage(512, layers[3])   # reference passes dilation=4: ignored there
        self.avgpool = nn.AvgPool2d(7)
        self.fc = nn.Linear(512, num_classes)
        self.register_buffer("_zero", torch.zeros(1), persistent=False)
        self.register_buffer("_one", torch.ones(1), persistent=False)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                fan = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / fan))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _stage(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self._inplanes != planes:
            if self._stride == self._output_stride:   # never true with output_stride 32
                self._dilation *= stride
                stride = 1
            else:
                self._stride *= stride
            downsample = nn.Sequential(nn.Conv2d(self._inplanes, planes, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes))
        seq = [BasicBlock(self._inplanes, planes, stride, downsample, dilation=self._dilation)]
        self._inplanes = planes
        seq += [BasicBlock(planes, planes, dilation=self._dilation) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def forward(self, x):
        x = self.maxpool(_bn_relu(self.bn1, self.conv1(x), None, self))
        x = self.layer2(self.layer1(x))
        x3 = self.layer3(x)
        return self.layer4(x3), x3


_PSP_MATRICES = {}


def _psp_matrices(sizes, h, w, device):
    """Fixed matrices of the pyramid: P (R, h*w) stacks the adaptive-average-pool operators of all bin sizes
    (bin i covers [floor(i*h/s), ceil((i+1)*h/s)), as the framework's AdaptiveAvgPool2d), U (h*w, R) the bilinear
    upsample operators back to (h, w) with align_corners=False (source index max(0, (o + 0.5) * s/h - 0.5), float32
    as the framework computes it); rows[k] = row range of bin size k; R = sum s*s."""
    key = (tuple(sizes), h, w, str(device))
    if key in _PSP_MATRICES:
        return _PSP_MATRICES[key]

    def pool_1d(s, n):
        m = torch.zeros(s, n, dtype=torch.float64)
        for i in range(s):
            a, b = (i * n) // s, -((-(i + 1) * n) // s)
            m[i, a:b] = 1.0 / (b - a)
        return m

    def up_1d(s, n):
        m = torch.zeros(n, s, dtype=torch.float32)
        scale = torch.tensor(float(s), dtype=torch.float32) / torch.tensor(float(n), dtype=torch.float32)
        for o in range(n):
            src = scale * (o + 0.5) - 0.5
            src = src.clamp(min=0.0)
            i0 = int(src)
            i1 = i0 + (1 if i0 < s - 1 else 0)
            l1 = src - i0
            m[o, i0] += 1.0 - l1
            m[o, i1] += l1
        return m

    ps, us, rows, r = [], [], [], 0
    for s_ in sizes:
        ps.append(torch.kron(pool_1d(s_, h), pool_1d(s_, w)).float())          # (s*s, h*w)
        us.append(torch.kron(up_1d(s_, h), up_1d(s_, w)))                      # (h*w, s*s)
        rows.append((r, r + s_ * s_))
        r += s_ * s_
    out = (torch.cat(ps, 0).to(device).contiguous(), torch.cat(us, 1).to(device).contiguous(), rows)
    _PSP_MATRICES[key] = out
    return out


_UNIT_CONSTS = {}


def _unit_consts(dev, c):
    """Constant blocks that turn the decoder's BatchNorm + PReLU passes into ReLU (+ mask): bn = (scale 1, shift 0, mean 0,
    invstd 1), bwdc = (1, 0, 0), slope 0."""
    key = (str(dev), c)
    if key not in _UNIT_CONSTS:
        bn = torch.zeros((4, c), dtype=torch.float32, device=dev)
        bn[0].fill_(1.0); bn[3].fill_(1.0)
        bwdc = torch.zeros((3, c), dtype=torch.float32, device=dev)
        bwdc[0].fill_(1.0)
        _UNIT_CONSTS[key] = (bn, bwdc, torch.zeros(1, dtype=torch.float32, device=dev))
    return _UNIT_CONSTS[key]


class _PSPLinearFn(torch.autograd.Function):
    """relu( x Wb_n^T + bias + U [ (P_k x) Ws_k^T Wb_k^T ]_k ) for x (B, HW, C) -- PSPModule's linear form (its docstring) as
    one autograd node with the backward written out.  Left to autograd, the column slices of the bottleneck weight and the row
    slices of the pooled map each came back as a zero-filled full-size gradient plus a copy plus an add (~60 framework
    launches per step for 10 % of the module's arithmetic); here every product writes its slice of the result in place
    (``out=`` on a view) and the bin-size blocks are contiguous because pooled maps are kept as (R, B, C).
    Plain tensor products (library GEMMs), any device / dtype: the float64 host test checks the algebra exactly."""

    @staticmethod
    def forward(ctx, x, pmat, umat, rows, mask, bias, wb, *ws):
        # mask: (B, Cout) Dropout2d factors applied after the ReLU (the drop_1 of Modified_PSPNet, model/modules.py:60) or None
        b, hw, c = x.shape
        cout, n, r = wb.shape[0], len(ws), pmat.shape[0]
        x2 = x.reshape(b * hw, c)
        pooled = torch.matmul(pmat, x).transpose(0, 1).contiguous()        # (R, B, C): a bin size = a contiguous row block
        mid = torch.empty_like(pooled)                                     # pooled_k Ws_k^T
        t = x.new_empty((r, b, cout))
        for k, (r0, r1) in enumerate(rows):
            torch.mm(pooled[r0:r1].view(-1, c), ws[k].t(), out=mid[r0:r1].view(-1, c))
            torch.mm(mid[r0:r1].view(-1, c), wb[:, k * c:(k + 1) * c].t(), out=t[r0:r1].view(-1, cout))
        acc = torch.addmm(bias, x2, wb[:, n * c:].t()).view(b, hw, cout)
        y = torch.baddbmm(acc, umat.unsqueeze(0).expand(b, -1, -1), t.transpose(0, 1))
        native = mask is not None and y.is_cuda and y.dtype == torch.float32 and cout % 4 == 0 and cout <= 1024
        if native:            # relu(y) * mask in one pass (istnet_nhwc_bn_prelu_apply with scale 1, shift 0, slope 0), in place
            from . import _native
            bn, _, zero = _unit_consts(y.device, cout)
            with torch.cuda.device(y.device):
                _native.check(_native.lib().istnet_nhwc_bn_prelu_apply(
                    b, hw, cout, y.data_ptr(), bn.data_ptr(), zero.data_ptr(), mask.data_ptr(), y.data_ptr(),
                    torch.cuda.current_stream(y.device).cuda_stream), "nhwc_bn_prelu_apply")
        else:
            torch.relu_(y)
            if mask is not None:
                y.mul_(mask.unsqueeze(1))
        ctx.save_for_backward(x, pmat, umat, wb, pooled, mid, y, mask if mask is not None else x.new_empty(0), *ws)
        ctx.rows, ctx.has_mask, ctx.native = rows, mask is not None, native
        return y

    @staticmethod
    def backward(ctx, dy):
        x, pmat, umat, wb, pooled, mid, y, mask, *ws = ctx.saved_tensors
        rows = ctx.rows
        b, hw, c = x.shape
        cout, n = wb.shape[0], len(ws)
        dy = dy.contiguous()
        if ctx.native:        # dy * mask where the (masked) output is positive: a dropped channel has no gradient either way
            from . import _native
            bn, bwdc, zero = _unit_consts(y.device, cout)
            g = torch.empty_like(y)
            with torch.cuda.device(y.device):
                _native.check(_native.lib().istnet_nhwc_bn_prelu_bwd_apply(
                    b, hw, cout, y.data_ptr(), dy.data_ptr(), bn.data_ptr(), bwdc.data_ptr(), zero.data_ptr(), mask.data_ptr(),
                    g.data_ptr(), torch.cuda.current_stream(y.device).cuda_stream), "nhwc_bn_prelu_bwd_apply")
        else:
            g = torch.ops.aten.threshold_backward(dy, y, 0)                # dy where y > 0
            if ctx.has_mask:
                g = g * mask.unsqueeze(1)
        g2, x2 = g.view(b * hw, cout), x.reshape(b * hw, c)
        dwb = torch.empty_like(wb)
        dx = torch.mm(g2, wb[:, n * c:]).view(b, hw, c)
        torch.mm(g2.t(), x2, out=dwb[:, n * c:])
        dbias = g2.sum(0)
        dt = torch.matmul(umat.t(), g).transpose(0, 1).contiguous()        # (R, B, Cout)
        dpooled = torch.empty_like(pooled)
        dws = []
        for k, (r0, r1) in enumerate(rows):
            dtk, wbk = dt[r0:r1].view(-1, cout), wb[:, k * c:(k + 1) * c]
            torch.mm(dtk.t(), mid[r0:r1].view(-1, c), out=dwb[:, k * c:(k + 1) * c])
            dmk = torch.mm(dtk, wbk)
            dws.append(torch.mm(dmk.t(), pooled[r0:r1].view(-1, c)))
            torch.mm(dmk, ws[k], out=dpooled[r0:r1].view(-1, c))
        dx.baddbmm_(pmat.t().unsqueeze(0).expand(b, -1, -1), dpooled.transpose(0, 1))
        return (dx, None, None, None, None, dbias, dwb, *dws)


class PSPModule(nn.Module):
    """Pyramid pooling at bin sizes (1,2,3,6) + 1x1 bottleneck.  [ref modules.py:10-34]

    Every operator between ``feats`` and the bottleneck's ReLU is linear, and a 1x1 convolution commutes with a spatial
    interpolation, so on the GPU (channels-last float32) the module is evaluated as

        relu( feats . Wf^T + bias  +  U . [ (P_k feats) . Ws_k^T . Wb_k^T ]_k )

    with the fixed pooling / upsampling matrices P, U of _psp_matrices: the bottleneck's slices Wb_k act on the s x s
    maps BEFORE the upsample, the 2560-channel concatenation never exists, the large product is 512 -> 1024 instead
    of 2560 -> 1024 (forward and both backward products), and pooling / upsampling are two small matrix products
    (deterministic: the framework's upsample backward uses atomics).  Same parameters, same state-dict keys."""

    def __init__(self, features, out_features=1024, sizes=(1, 2, 3, 6)):
        super().__init__()
        self.stages = nn.ModuleList([
            nn.Sequential(nn.AdaptiveAvgPool2d(output_size=(size, size)),
                          nn.Conv2d(features, features, kernel_size=1, bias=False)) for size in sizes])
        self.bottleneck = nn.Conv2d(features * (len(sizes) + 1), out_features, kernel_size=1)
        self.relu = nn.ReLU()
        self.sizes = tuple(sizes)

    def _forward_linear(self, feats, mask=None):
        b, c, h, w = feats.shape
        cout = self.bottleneck.out_channels
        x = feats.permute(0, 2, 3, 1).reshape(b, h * w, c)                 # a view of the channels-last map
        pmat, umat, rows = _psp_matrices(self.sizes, h, w, feats.device)
        pmat, umat = pmat.to(x.dtype), umat.to(x.dtype)                   # no-ops for float32
        y = _PSPLinearFn.apply(x, pmat, umat, tuple(rows), mask, self.bottleneck.bias,
                               self.bottleneck.weight.view(cout, (len(self.stages) + 1) * c),
                               *[st[1].weight.view(c, c) for st in self.stages])
        return y.view(b, h, w, cout).permute(0, 3, 1, 2)                   # (B, Cout, h, w), channels-last

    def forward(self, feats, drop=None):
        """``drop``: the nn.Dropout2d the caller applies to the module's output (reference model/modules.py:60), folded into
        the ReLU pass of the linear form; otherwise applied here."""
        if (USE_FUSED and feats.is_cuda and feats.dtype == torch.float32 and feats.dim() == 4
                and feats.is_contiguous(memory_format=torch.channels_last) and self.bottleneck.bias is not None):
            mask = None
            if drop is not None and drop.training and drop.p > 0:
                keep = 1.0 - drop.p
                mask = torch.empty((feats.shape[0], self.bottleneck.out_channels), dtype=torch.float32,
                                   device=feats.device).bernoulli_(keep).div_(keep)
            return self._forward_linear(feats, mask)
        h, w = feats.size(2), feats.size(3)
        priors = [F.interpolate(stage(feats), size=(h, w), mode="bilinear", align_corners=False)
                  for stage in self.stages] + [feats]
        out = self.relu(self.bottleneck(torch.cat(priors, 1)))
        return drop(out) if drop is not None else out


UPCONV_MIN_CIN = 64          # all three decoder stages; below this moving q (9 x Cout channels) costs more than it saves


def _upconv_tail_forward(q, bias, cout):
    from . import _native
    b, h, w, _ = q.shape
    y = torch.empty((b, cout, 2 * h, 2 * w), dtype=q.dtype, device=q.device, memory_format=torch.channels_last)
    with torch.cuda.device(q.device):
        _native.check(_native.lib().istnet_upconv3_fwd_nhwc(
            b, cout, h, w, 2 * h, 2 * w, q.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
            torch.cuda.current_stream(q.device).cuda_stream), "upconv3_fwd_nhwc")
    return y


def _upconv_tail_backward(dy, dims):
    from . import _native
    b, h, w, cout = dims
    dy = dy.contiguous(memory_format=torch.channels_last)
    dq = torch.empty((b, h, w, 9 * cout), dtype=dy.dtype, device=dy.device)
    with torch.cuda.device(dy.device):
        _native.check(_native.lib().istnet_upconv3_bwd_nhwc(
            b, cout, h, w, 2 * h, 2 * w, dy.data_ptr(), dq.data_ptr(),
            torch.cuda.current_stream(dy.device).cuda_stream), "upconv3_bwd_nhwc")
    return dq


class _UpConvTailFn(torch.autograd.Function):
    """q (B, h, w, 9*Cout) -> y (B, Cout, 2h, 2w) channels-last: the full-size part of upsample -> conv3x3
    (include/istnet_rgb.h, istnet_upconv3_*).  Linear in q; the bias gradient is the plain sum of dy."""

    @staticmethod
    def forward(ctx, q, bias, cout):
        b, h, w, _ = q.shape
        ctx.dims = (b, h, w, cout)
        ctx.has_bias = bias is not None
        return _upconv_tail_forward(q, bias, cout)

    @staticmethod
    def backward(ctx, dy):
        dq = _upconv_tail_backward(dy, ctx.dims)
        dbias = dy.sum(dim=(0, 2, 3)) if ctx.has_bias and ctx.needs_input_grad[1] else None
        return dq, dbias, None


class _PointMixFn(torch.autograd.Function):
    """q = x @ wr for x (points, Cin) with very many points and a small (Cin, N) weight.  Forward and the input
    gradient are plain products (hipBLASLt runs them at 100-145 TFLOP/s); the weight gradient x^T dq reduces over the
    points, has only Cin x N outputs, and the library gives it too few workgroups (64 / 33 TFLOP/s on the up_1 / up_2
    shapes): it is issued as a batch of 16 / 32 products over slices of the points and summed in a fixed order
    (132 / 113 TFLOP/s, tools/exp/gemm_lib.py)."""

    @staticmethod
    def forward(ctx, x, wr):
        ctx.save_for_backward(x, wr)
        return torch.matmul(x, wr)

    @staticmethod
    def backward(ctx, dq):
        x, wr = ctx.saved_tensors
        dq = dq.contiguous()
        dx = torch.matmul(dq, wr.t()) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            m = x.shape[0]
            s = 32 if m >= 200000 else 16
            if m % s == 0 and m // s >= 256 and x.is_contiguous():
                dw = torch.bmm(x.view(s, m // s, x.shape[1]).transpose(1, 2), dq.view(s, m // s, dq.shape[1])).sum(0)
            else:
                dw = torch.matmul(x.t(), dq)
        return dx, dw




def _bn_prelu_forward(y, gamma, beta, slope, mask, running_mean, running_var, momentum, eps):
    """-> (z, bn): statistics pass, finalize (running statistics updated, momentum from its device slot), apply pass."""
    from . import _native
    lib = _native.lib()
    b, c, h, w = y.shape
    dev = y.device
    rows = b * h * w
    nparts = lib.istnet_nhwc_stat_parts(rows)
    part = torch.empty((2, c, nparts), dtype=torch.float32, device=dev)
    bn = torch.empty((4, c), dtype=torch.float32, device=dev)
    z = torch.empty_like(y, memory_format=torch.channels_last)
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream(dev).cuda_stream
        _native.check(lib.istnet_nhwc_channel_stats(rows, c, y.data_ptr(), part[0].data_ptr(), part[1].data_ptr(), st),
                      "nhwc_channel_stats")
        _native.check(lib.istnet_bn_finalize_fwd(
            c, nparts, float(rows), part[0].data_ptr(), part[1].data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps),
            momentum, running_mean.data_ptr() if running_mean is not None else None,
            running_var.data_ptr() if running_var is not None else None, bn.data_ptr(), st), "bn_finalize_fwd")
        _native.check(lib.istnet_nhwc_bn_prelu_apply(b, h * w, c, y.data_ptr(), bn.data_ptr(), slope.data_ptr(),
                                                     mask.data_ptr() if mask is not None else None, z.data_ptr(), st),
                      "nhwc_bn_prelu_apply")
    return z, bn


def _bn_prelu_backward(y, dz, gamma, slope, bn, mask, want_colsum):
    """-> (dy, dgamma, dbeta, dslope, colsum): statistics pass, finalize (which also leaves the slope gradient and, when
    asked for, the column sums of dy = the preceding convolution's bias gradient), apply pass."""
    from . import _native
    lib = _native.lib()
    b, c, h, w = y.shape
    dev = y.device
    rows = b * h * w
    dz = dz.contiguous(memory_format=torch.channels_last)
    nparts = lib.istnet_nhwc_stat_parts(rows)
    f32 = dict(dtype=torch.float32, device=dev)
    part, pslope = torch.empty((2, c, nparts), **f32), torch.empty((nparts,), **f32)
    dgamma, dbeta, bwdc = torch.empty((c,), **f32), torch.empty((c,), **f32), torch.empty((3, c), **f32)
    dslope = torch.empty(slope.shape, **f32)
    colsum = torch.empty((c,), **f32) if want_colsum else None
    dy = torch.empty_like(y, memory_format=torch.channels_last)
    mptr = mask.data_ptr() if mask is not None else None
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream(dev).cuda_stream
        _native.check(lib.istnet_nhwc_bn_prelu_bwd_stats(b, h * w, c, y.data_ptr(), dz.data_ptr(), bn.data_ptr(),
                                                         slope.data_ptr(), mptr, part[0].data_ptr(), part[1].data_ptr(),
                                                         pslope.data_ptr(), st), "nhwc_bn_prelu_bwd_stats")
        _native.check(lib.istnet_nhwc_bn_prelu_bwd_finalize(
            c, nparts, float(rows), part[0].data_ptr(), part[1].data_ptr(), pslope.data_ptr(), gamma.data_ptr(), bn.data_ptr(),
            dgamma.data_ptr(), dbeta.data_ptr(), bwdc.data_ptr(), colsum.data_ptr() if colsum is not None else None,
            dslope.data_ptr(), st), "nhwc_bn_prelu_bwd_finalize")
        _native.check(lib.istnet_nhwc_bn_prelu_bwd_apply(b, h * w, c, y.data_ptr(), dz.data_ptr(), bn.data_ptr(),
                                                         bwdc.data_ptr(), slope.data_ptr(), mptr, dy.data_ptr(), st),
                      "nhwc_bn_prelu_bwd_apply")
    return dy, dgamma, dbeta, dslope, colsum


class _BnPReLUDropFn(torch.autograd.Function):
    """BatchNorm2d (training statistics) -> PReLU (one slope) [-> Dropout2d mask] of a channels-last map in two streaming
    passes per direction (include/istnet_rgb.h, istnet_nhwc_*; finalizes of include/istnet_pw.h): the tail of a decoder
    stage (reference model/modules.py:25-34,63-65).  The framework's sequence -- MIOpen's two-kernel BatchNorm, a PReLU
    kernel, a dropout multiply -- moves the 75-302 MB map 7 times forward and ~10 times backward; this node 3 and 5 times.
    Statistics as in the point branch: fp32 partial sums per 32-512 pixels, combined in float64; running statistics updated
    with torch's semantics (momentum, unbiased variance).  mask: (B, C) Dropout2d factors (0 or 1 / (1 - p)) or None."""

    @staticmethod
    def forward(ctx, y, gamma, beta, slope, mask, running_mean, running_var, momentum, eps):
        z, bn = _bn_prelu_forward(y, gamma, beta, slope, mask, running_mean, running_var, momentum, eps)
        ctx.save_for_backward(y, gamma, slope, bn, mask if mask is not None else torch.empty(0, device=y.device))
        ctx.has_mask = mask is not None
        return z

    @staticmethod
    def backward(ctx, dz):
        y, gamma, slope, bn, mask = ctx.saved_tensors
        dy, dgamma, dbeta, dslope, _ = _bn_prelu_backward(y, dz, gamma, slope, bn, mask if ctx.has_mask else None, False)
        return dy, dgamma, dbeta, dslope, None, None, None, None, None


class _UpConvNormFn(torch.autograd.Function):
    """_UpConvTailFn followed by _BnPReLUDropFn as ONE node: the tail of a decoder stage from the channel-mixed small map q
    to the stage's output.  Being one node, the convolution's bias gradient (the column sums of the gradient that leaves the
    normalisation) is taken from the backward statistics instead of a pass over the full-size map."""

    @staticmethod
    def forward(ctx, q, bias, cout, gamma, beta, slope, mask, running_mean, running_var, momentum, eps):
        b, h, w, _ = q.shape
        ctx.dims = (b, h, w, cout)
        ctx.has_mask, ctx.has_bias = mask is not None, bias is not None
        y = _upconv_tail_forward(q, bias, cout)
        z, bn = _bn_prelu_forward(y, gamma, beta, slope, mask, running_mean, running_var, momentum, eps)
        ctx.save_for_backward(y, gamma, slope, bn, mask if mask is not None else torch.empty(0, device=q.device))
        return z

    @staticmethod
    def backward(ctx, dz):
        y, gamma, slope, bn, mask = ctx.saved_tensors
        want = ctx.has_bias and ctx.needs_input_grad[1]
        dy, dgamma, dbeta, dslope, colsum = _bn_prelu_backward(y, dz, gamma, slope, bn, mask if ctx.has_mask else None, want)
        return (_upconv_tail_backward(dy, ctx.dims), colsum, None, dgamma, dbeta, dslope, None, None, None, None, None)


class PSPUpsample(nn.Module):
    """Upsample(2x, bilinear, align_corners) -> Conv2d(3x3, padding 1) -> BatchNorm2d -> PReLU  [ref modules.py:36-49].

    On the GPU (channels-last float32, Cin >= UPCONV_MIN_CIN) the convolution is not run at full size: by linearity
    conv3x3(U p) = sum_taps shift_tap(U (W_tap p)), so the channel mixing is ONE matrix product on the small map,
    q = p (B h w, Cin) x Wr (Cin, 9 Cout) -- a quarter of the convolution's flops, and the same for both backward
    products, which autograd takes from the matmul -- and an interpolate / shift / add kernel produces the full-size
    map.  up_1 (1024 -> 256 at 48 x 48) is 348 GFLOP per pass as a convolution, 87 this way.  Same parameters and
    state-dict keys (conv.1.weight / conv.1.bias)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Sequential(Upsample2x(), nn.Conv2d(in_channels, out_channels, 3, padding=1),
                                  nn.BatchNorm2d(out_channels), PReLU())

    def _split_ok(self, x):
        conv = self.conv[1]
        return (USE_FUSED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
                and x.is_contiguous(memory_format=torch.channels_last) and x.shape[2] > 1 and x.shape[3] > 1
                and conv.in_channels >= UPCONV_MIN_CIN and conv.out_channels % 4 == 0)

    def forward(self, x, drop=None):
        """``drop``: the nn.Dropout2d the caller applies to this stage's output (reference modules.py:63-65), fused into
        the stage's tail when it can be; otherwise applied here."""
        if not self._split_ok(x):
            out = self.conv(x)
            return drop(out) if drop is not None else out
        conv = self.conv[1]
        b, cin, h, w = x.shape
        cout = conv.out_channels
        wr = conv.weight.permute(1, 2, 3, 0).reshape(cin, 9 * cout)          # Wr[ci][(ky*3+kx)*Cout + co]
        q = _PointMixFn.apply(x.permute(0, 2, 3, 1).reshape(b * h * w, cin), wr).view(b, h, w, 9 * cout)
        bn, act = self.conv[2], self.conv[3]
        if (USE_FUSED and bn.training and bn.affine and bn.momentum is not None and act.weight.numel() == 1
                and cout % 4 == 0 and cout <= 1024 and torch.is_grad_enabled()):
            mask = None
            if drop is not None and drop.training and drop.p > 0:
                keep = 1.0 - drop.p
                mask = torch.empty((b, cout), dtype=torch.float32, device=x.device).bernoulli_(keep).div_(keep)
            _bump_batch_counter(bn)
            return _UpConvNormFn.apply(q, conv.bias, cout, bn.weight, bn.bias, act.weight, mask,
                                       bn.running_mean if bn.track_running_stats else None,
                                       bn.running_var if bn.track_running_stats else None, bn_momentum_ptr(bn, x.device), bn.eps)
        y = _UpConvTailFn.apply(q, conv.bias, cout)
        out = act(bn(y))
        return drop(out) if drop is not None else out






def _moments(rows):
    """sum_p u_p (C) and sum_p u_p u_p^T (C, C) of rows (P, C).  CUDA, C = 64 (the decoder's last stage), contiguous rows:
    one streaming pass (istnet_nhwc_gram64: fp32-MFMA Gram partials per workgroup, summed in float64 in a fixed order;
    returns float64).  Otherwise the reduction over P as a batch of slice products (the library under-fills the chip on a
    (C, P) x (P, C) product, see _PointMixFn), summed in a fixed order."""
    p, c = rows.shape
    if (USE_FUSED and rows.is_cuda and rows.dtype == torch.float32 and c == 64 and rows.is_contiguous()
            and rows.data_ptr() % 16 == 0):
        from . import _native
        lib = _native.lib()
        nparts = lib.istnet_nhwc_gram64_parts(p)
        part2 = torch.empty((nparts, 64, 64), dtype=torch.float32, device=rows.device)
        part1 = torch.empty((nparts, 64), dtype=torch.float32, device=rows.device)
        s2 = torch.empty((64, 64), dtype=torch.float64, device=rows.device)
        s1 = torch.empty((64,), dtype=torch.float64, device=rows.device)
        with torch.cuda.device(rows.device):
            _native.check(lib.istnet_nhwc_gram64(p, rows.data_ptr(), part2.data_ptr(), part1.data_ptr(), s2.data_ptr(),
                                                 s1.data_ptr(), torch.cuda.current_stream(rows.device).cuda_stream),
                          "nhwc_gram64")
        return s1, s2
    s = 32 if p >= 200000 else 16
    if p % s == 0 and p // s >= 256:
        v = rows.view(s, p // s, c)
        return v.sum(1).sum(0), torch.bmm(v.transpose(1, 2), v).sum(0)
    return rows.sum(0), torch.matmul(rows.t(), rows)


                                    # instead of ~85 framework launches of small float64 algebra


def _final_native_ok(u, choose, weight, slope):
    return (USE_FUSED and u.is_cuda and u.dtype == torch.float32 and u.shape[1] == 64
            and u.is_contiguous(memory_format=torch.channels_last) and u.data_ptr() % 16 == 0 and weight.shape[0] <= 512
            and slope.numel() == 1 and choose.dtype == torch.int64)


class _FinalAtChosenNativeFn(torch.autograd.Function):
    """_FinalAtChosenFn (below: the derivation and the framework form) as two calls into include/istnet_rgb.h:
    istnet_final_chosen_forward (moments -> batch statistics -> y, zhat at the chosen pixels) and
    istnet_final_chosen_backward (sums -> constants -> dense affine pass -> direct path -> parameter gradients)."""

    @staticmethod
    def forward(ctx, u, choose, weight, bias, gamma, beta, slope, running_mean, running_var, momentum_ptr, eps):
        from . import _native
        lib = _native.lib()
        b, c, h, w = u.shape
        n, cout, dev = choose.shape[1], weight.shape[0], u.device
        choose = choose.contiguous()
        w2 = weight.reshape(cout, c).contiguous()
        nparts = lib.istnet_nhwc_gram64_parts(b * h * w)
        part2 = torch.empty((nparts, 64, 64), dtype=torch.float32, device=dev)
        part1 = torch.empty((nparts, 64), dtype=torch.float32, device=dev)
        s2 = torch.empty((64, 64), dtype=torch.float64, device=dev)
        s1 = torch.empty((64,), dtype=torch.float64, device=dev)
        stat = torch.empty((3, cout), dtype=torch.float64, device=dev)
        y = torch.empty((b, cout, n), dtype=torch.float32, device=dev)
        zhat = torch.empty((b, cout, n), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _native.check(lib.istnet_final_chosen_forward(
                b, h * w, n, cout, u.data_ptr(), choose.data_ptr(), w2.data_ptr(), bias.data_ptr(), gamma.data_ptr(),
                beta.data_ptr(), slope.data_ptr(), running_mean.data_ptr() if running_mean is not None else 0,
                running_var.data_ptr() if running_mean is not None else 0, momentum_ptr if running_mean is not None else 0,
                float(eps), part2.data_ptr(), part1.data_ptr(), s2.data_ptr(), s1.data_ptr(), stat.data_ptr(), y.data_ptr(),
                zhat.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "final_chosen_forward")
        ctx.save_for_backward(u, choose, w2, bias, gamma, beta, slope, s1, s2, stat, zhat)
        ctx.wshape = weight.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import _native
        lib = _native.lib()
        u, choose, w2, bias, gamma, beta, slope, s1, s2, stat, zhat = ctx.saved_tensors
        b, c, h, w = u.shape
        n, cout, dev = choose.shape[1], w2.shape[0], u.device
        dy = dy.contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        part = torch.empty((3, cout, b), **f32)
        bwdc = torch.empty((4, cout), dtype=torch.float64, device=dev)
        amat, c0 = torch.empty((64, 64), **f32), torch.empty((64,), **f32)
        dwp = torch.empty((lib.istnet_final_chosen_workgroups(b * n), cout, 64), **f32)
        du = torch.empty((b, h, w, c), **f32)
        dw, db = torch.empty((cout, c), **f32), torch.empty((cout,), **f32)
        dgamma, dbeta, dslope = torch.empty((cout,), **f32), torch.empty((cout,), **f32), torch.empty(slope.shape, **f32)
        with torch.cuda.device(dev):
            _native.check(lib.istnet_final_chosen_backward(
                b, h * w, n, cout, u.data_ptr(), choose.data_ptr(), w2.data_ptr(), bias.data_ptr(), gamma.data_ptr(),
                beta.data_ptr(), slope.data_ptr(), s2.data_ptr(), s1.data_ptr(), stat.data_ptr(), dy.data_ptr(), zhat.data_ptr(),
                part.data_ptr(), bwdc.data_ptr(), amat.data_ptr(), c0.data_ptr(), dwp.data_ptr(), du.data_ptr(), dw.data_ptr(),
                db.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), dslope.data_ptr(),
                torch.cuda.current_stream(dev).cuda_stream), "final_chosen_backward")
        return (du.permute(0, 3, 1, 2), None, dw.view(ctx.wshape), db, dgamma, dbeta, dslope, None, None, None, None)


class _FinalAtChosenFn(torch.autograd.Function):
    """`final` = Conv2d(1x1) -> BatchNorm2d (train mode) -> PReLU of the decoder, evaluated at the chosen pixels only.

    IST-Net reads N of the H*W pixels of the (B, 128, H, W) feature map (ist_net.py:41-45).  In training mode the
    BatchNorm statistics are taken over ALL pixels, but z = W u + b is linear in the layer's input u, so they follow
    exactly from the first and second moments of u (one pass over u, a (C, P) x (P, C) product):
        mean_z = W mean_u + b,   var_z[c] = w_c^T Cov(u) w_c.
    The map itself (604 MB at B = 32) and the conv / BatchNorm / PReLU passes over it, forward and backward, are never
    run.  Backward: the loss reaches u through the chosen pixels and through the two statistics; the second path is
    affine in u for every pixel, dL/du_p = A u_p + c0 with A = W^T diag(k) W (one dense (P, C) x (C, C) product), and
    the parameter gradients need only the moments.  Same values as the dense composition up to summation order;
    running statistics updated as BatchNorm2d does (momentum, unbiased variance)."""

    @staticmethod
    def forward(ctx, u, choose, weight, bias, gamma, beta, slope, running_mean, running_var, momentum, eps):
        b, c, h, w = u.shape
        npix = b * h * w
        rows = u.permute(0, 2, 3, 1).reshape(npix, c)                      # view of the channels-last map
        s1, s2 = _moments(rows)
        w2 = weight.reshape(weight.shape[0], c)
        wd, bd = w2.double(), bias.double()
        m_u = s1.double() / npix
        cov = s2.double() / npix - torch.outer(m_u, m_u)
        mu = wd @ m_u + bd
        var = ((wd @ cov) * wd).sum(1).clamp_(min=0.0)
        istd = (var + eps).rsqrt()
        n = choose.shape[1]
        u_sel = torch.gather(rows.view(b, h * w, c), 1, choose.unsqueeze(-1).expand(-1, -1, c))     # (B, N, C)
        z = torch.addmm(bias, u_sel.reshape(b * n, c), w2.t())                                      # (B N, Cout)
        zhat = (z - mu.to(z.dtype)) * istd.to(z.dtype)
        v = zhat * gamma + beta
        y = torch.where(v > 0, v, v * slope)
        if running_mean is not None:
            with torch.no_grad():
                # momentum: the module's one-element device slot (pytorch_utils.bn_momentum_tensor), so a captured step
                # follows BNMomentumScheduler.step; lerp(start, end, w) = (1 - w) start + w end
                running_mean.lerp_(mu.to(running_mean.dtype), momentum)
                running_var.lerp_((var * (npix / max(npix - 1, 1))).to(running_var.dtype), momentum)
        ctx.save_for_backward(u, choose, weight, bias, gamma, slope, u_sel, zhat, v, s1, s2, mu, istd)
        return y.view(b, n, -1).transpose(1, 2).contiguous()                                        # (B, Cout, N)

    @staticmethod
    def backward(ctx, dy):
        u, choose, weight, bias, gamma, slope, u_sel, zhat, v, s1, s2, mu, istd = ctx.saved_tensors
        b, c, h, w = u.shape
        npix, n, cout = b * h * w, choose.shape[1], weight.shape[0]
        w2 = weight.reshape(cout, c)
        g = dy.transpose(1, 2).reshape(b * n, cout)
        neg = v <= 0
        dslope = (g * v * neg).sum().reshape(slope.shape)
        gv = torch.where(neg, g * slope, g)
        dbeta, dgamma = gv.sum(0), (gv * zhat).sum(0)
        gz = gv * gamma                                                    # dL/dzhat at the chosen pixels
        istd_f = istd.to(gz.dtype)
        # z - mu at the chosen pixels = zhat / istd; the statistics couple every pixel: dL/dz_p += a + k (z_p - mu)
        a = (-(istd * gz.sum(0).double()) / npix)                          # dL/dmu / P                     (float64, Cout)
        k = (-(istd ** 2) * (gz * zhat).sum(0).double() / npix)            # 2 dL/dvar / P = -istd^2 sum(gz zhat) / P
        dz = gz * istd_f                                                   # direct path, (B N, Cout)
        wd = w2.double()
        off = a + k * (bias.double() - mu)                                 # constant part of the dense term
        amat = (wd.t() * k) @ wd                                           # W^T diag(k) W  (C, C)
        c0 = wd.t() @ off
        rows = u.permute(0, 2, 3, 1).reshape(npix, c)
        if (USE_FUSED and rows.is_cuda and rows.dtype == torch.float32 and c == 64 and rows.is_contiguous()
                and rows.data_ptr() % 16 == 0):
            from . import _native
            du = torch.empty_like(rows)
            a32, c32 = amat.to(torch.float32).contiguous(), c0.to(torch.float32).contiguous()
            with torch.cuda.device(rows.device):
                _native.check(_native.lib().istnet_nhwc_rowmix64(
                    npix, rows.data_ptr(), a32.data_ptr(), c32.data_ptr(), du.data_ptr(),
                    torch.cuda.current_stream(rows.device).cuda_stream), "nhwc_rowmix64")
        else:
            du = torch.addmm(c0.to(rows.dtype), rows, amat.to(rows.dtype).t())           # dense: A u_p + c0, (P, C)
        du = du.view(b, h * w, c)
        du.scatter_add_(1, choose.unsqueeze(-1).expand(-1, -1, c), torch.matmul(dz, w2).view(b, n, c))
        du = du.view(b, h, w, c).permute(0, 3, 1, 2)                       # (B, C, H, W), channels-last
        s1d, s2d = s1.double(), s2.double()
        dw = (dz.t() @ u_sel.reshape(b * n, c)).double() + torch.outer(a, s1d) \
            + k.unsqueeze(1) * (wd @ s2d + torch.outer(bias.double() - mu, s1d))
        db = dz.sum(0).double() + npix * a + k * (wd @ s1d + npix * (bias.double() - mu))
        return (du, None, dw.to(weight.dtype).view_as(weight), db.to(bias.dtype), dgamma, dbeta, dslope, None, None, None, None)


class Modified_PSPNet(nn.Module):
    """[ref modules.py:51-81]"""

    def __init__(self, sizes=(1, 2, 3, 6), psp_size=512):
        super().__init__()
        self.feats = ResNet((2, 2, 2, 2))
        self.psp = PSPModule(psp_size, 1024, sizes)
        self.drop_1 = nn.Dropout2d(p=0.3)
        self.up_1 = PSPUpsample(1024, 256)
        self.up_2 = PSPUpsample(256, 64)
        self.up_3 = PSPUpsample(64, 64)
        self.drop_2 = nn.Dropout2d(p=0.15)
        self.final = nn.Sequential(nn.Conv2d(64, 128, kernel_size=1), nn.BatchNorm2d(128), PReLU())

    def forward(self, x, choose=None):
        """rgb (B,3,H,W) -> (B,128,H,W); with ``choose`` (B,N) flat pixel indices -> (B,128,N), the features of the chosen
        pixels only: eval mode through ``_tail_at``, training mode through ``_FinalAtChosenFn`` (exact batch statistics
        from the moments of the last stage's input)."""
        f, _ = self.feats(x)
        p = self.psp(f, drop=self.drop_1)
        p = self.up_1(p, drop=self.drop_2)
        p = self.up_2(p, drop=self.drop_2)
        if choose is not None and not self.training:
            return self._tail_at(p, choose)
        u = self.up_3(p)
        if choose is not None and self._train_gather_ok(u):
            conv, bn, act = self.final[0], self.final[1], self.final[2]
            _bump_batch_counter(bn)
            rm = bn.running_mean if bn.track_running_stats else None
            rv = bn.running_var if bn.track_running_stats else None
            if _final_native_ok(u, choose, conv.weight, act.weight):
                return _FinalAtChosenNativeFn.apply(u, choose, conv.weight, conv.bias, bn.weight, bn.bias, act.weight, rm, rv,
                                                    bn_momentum_ptr(bn, u.device), bn.eps)
            return _FinalAtChosenFn.apply(u, choose, conv.weight, conv.bias, bn.weight, bn.bias, act.weight, rm, rv,
                                          bn_momentum_tensor(bn, u.device), bn.eps)
        out = self.final(u)
        if choose is None:
            return out
        b, d = out.size(0), out.size(1)                                     # dense map, then the gather (reference order)
        if not out.is_contiguous() and out.is_contiguous(memory_format=torch.channels_last):
            rows = out.permute(0, 2, 3, 1).reshape(b, -1, d)
            return torch.gather(rows, 1, choose.unsqueeze(-1).expand(-1, -1, d)).transpose(1, 2).contiguous()
        return torch.gather(out.reshape(b, d, -1), 2, choose.unsqueeze(1).expand(-1, d, -1)).contiguous()

    def _train_gather_ok(self, u):
        conv, bn, act = self.final[0], self.final[1], self.final[2]
        return (USE_FUSED and self.training and bn.training and u.is_cuda and u.dtype == torch.float32
                and u.is_contiguous(memory_format=torch.channels_last) and conv.bias is not None and bn.affine
                and bn.momentum is not None and act.weight.numel() == 1)

    def _tail_at(self, p, choose):
        """Eval mode: the last decoder stage (`up_3`: 2x bilinear upsample, 3x3 conv, BatchNorm, PReLU) and `final`
        (1x1 conv, BatchNorm, PReLU) evaluated at the chosen pixels only.  IST-Net reads N of the H*W output pixels
        (ist_net.py:41-45) and every op after the upsample is local to a 3x3 window of it, so the 3x3 conv becomes a
        (B*N, 9*64) x (9*64, 64) product over gathered windows instead of a convolution over the full 192x192 map, and
        the (B,64,H,W) conv output and the (B,128,H,W) feature map (1.2 GB at B=64) are never materialised.  Same
        arithmetic per pixel as the dense path up to summation order; not valid in training mode, where BatchNorm
        takes batch statistics over all pixels."""
        upsample, conv, bn, act = self.up_3.conv[0], self.up_3.conv[1], self.up_3.conv[2], self.up_3.conv[3]
        up = upsample(p)                                             # (B, C, H, W)
        b, c, h, w = up.shape
        if not up.is_contiguous() and up.is_contiguous(memory_format=torch.channels_last):
            rows = up.permute(0, 2, 3, 1).reshape(b, h * w, c)       # view
        else:
            rows = up.reshape(b, c, h * w).transpose(1, 2)           # strided view; gather handles it
        r, q = choose // w, choose % w                               # (B, N)
        d = torch.arange(-1, 2, device=choose.device)               # (a kernel, not a host copy: capturable)
        rr = (r.unsqueeze(-1) + d.repeat_interleave(3)).clamp_(0, h - 1)     # (B, N, 9): window rows, kernel-row major
        qq = (q.unsqueeze(-1) + d.repeat(3)).clamp_(0, w - 1)
        inside = ((r.unsqueeze(-1) + d.repeat_interleave(3) == rr) & (q.unsqueeze(-1) + d.repeat(3) == qq))
        flat = (rr * w + qq).reshape(b, -1)                          # (B, N*9)
        win = torch.gather(rows, 1, flat.unsqueeze(-1).expand(-1, -1, c))    # (B, N*9, C)
        win = (win * inside.reshape(b, -1, 1)).reshape(b, choose.size(1), 9 * c)   # zero padding of the conv
        w2 = conv.weight.permute(0, 2, 3, 1).reshape(conv.out_channels, 9 * c)      # (Cout, kr, kc, Cin)
        y = torch.matmul(win, w2.t()) + conv.bias                    # (B, N, Cout)
        y = F.batch_norm(y.transpose(1, 2), bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
        y = F.prelu(y, act.weight)                                   # (B, Cout, N)
        conv, bn, act = self.final[0], self.final[1], self.final[2]
        y = F.conv1d(y, conv.weight.view(conv.out_channels, -1, 1), conv.bias)
        y = F.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
        return F.prelu(y, act.weight).contiguous()


class ModifiedResnet(nn.Module):
    """`rgb_cam_extractor` of IST-Net.  [ref modules.py:232-241]"""
    gathers_choose = True     # forward(x, choose) returns the chosen pixels' features in eval mode

    def __init__(self):
        super().__init__()
        self.model = Modified_PSPNet(sizes=(1, 2, 3, 6), psp_size=512)

    def forward(self, x, choose=None):
        global _COUNTER_SCOPE
        outer, _COUNTER_SCOPE = _COUNTER_SCOPE, []
        try:
            return self.model(x, choose)
        finally:
            pending, _COUNTER_SCOPE = _COUNTER_SCOPE, outer
            if pending:
                torch._foreach_add_(pending, 1)
